"""Multi-rank host logic on CPU: gloo, world_size 2 (the RCCL data path itself needs GPUs)."""

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT
from pysteps_amd import parallel


def test_partition_is_disjoint_and_complete():
    for n, w in [(48, 8), (48, 5), (3, 8), (0, 4), (7, 1)]:
        seen = []
        for r in range(w):
            part = parallel.partition(n, w, r)
            seen += list(part)
            assert all(parallel.owner_of(j, n, w) == r for j in part)
        assert seen == list(range(n))
    assert list(parallel.partition(48, 8, 3)) == list(range(18, 24))  # 6 members per GPU
    with pytest.raises(ValueError):
        parallel.partition(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_control_plane(tmp_path):
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path),
    ]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [r["max"] for r in res] == [2.0, 2.0]          # max over ranks seen by both
    assert [r["token_len"] for r in res] == [128, 128]    # rank-0 payload reached rank 1
    assert res[0]["mine"] == list(range(24)) and res[1]["mine"] == list(range(24, 48))
    assert res[0]["owners"] == [0] * 24 and res[1]["owners"] == [1] * 24
    # config 5: the ranks' row bands tile the 8192 rows, the default plan has no collective in the step
    assert [tuple(r["config5"]["rows"]) for r in res] == [(0, 4096), (4096, 8192)]
    assert all(r["config5"]["lk"] == "replicated" and r["config5"]["collectives_per_step"] == 0 for r in res)
    assert parallel.config5_plan(8192, 8, 3, lk="banded")["collectives_per_step"] == 5
    # the member shards of bench.py's N > 1 leg (6 per GPU): disjoint, complete, and the perturbators
    # each rank derives from the ensemble seed are the slices of ONE seed chain
    from pysteps_amd.extrapolation.ensemble import steps_perturbators

    assert res[0]["shard"] == list(range(6)) and res[1]["shard"] == list(range(6, 12))
    chain = [p["eps_par"] for p in steps_perturbators(12, 42, 1.0, 5.0)]
    assert res[0]["eps_par"] + res[1]["eps_par"] == chain and len(set(chain)) == 12
    # ... and so are the members' random streams (the member loop of the N > 1 bench leg draws from them)
    from pysteps_amd.extrapolation.ensemble import steps_noise_generators

    draws = [float(rs.standard_normal()) for rs in steps_noise_generators(12, 42)]
    assert res[0]["first_draws"] + res[1]["first_draws"] == draws and len(set(draws)) == 12


def test_communicator_needs_gpu():
    from pysteps_amd import _lib

    if _lib.load().psh_init(-1) == 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.HipLibraryError):
        parallel.Communicator(0, 1, lambda payload: payload)
