"""RCCL communicator on one GPU (world_size 1): dlopen, ncclCommInitRank, broadcast, all-gather."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_roundtrip():
    from pysteps_amd import parallel
    from pysteps_amd.device import DeviceArray, synchronize

    comm = parallel.Communicator(0, 1, lambda payload: payload)
    try:
        a = np.arange(1 << 16, dtype=np.float32).reshape(2, 128, 256)
        d = DeviceArray.from_host(a)
        comm.broadcast(d, root=0)
        g = comm.allgather(d)
        synchronize()
        assert np.array_equal(d.to_host(), a)
        assert g.shape == (1, 2, 128, 256) and np.array_equal(g.to_host()[0], a)
    finally:
        comm.close()


def test_sharded_extrapolate_single_rank():
    from pysteps_amd import parallel
    from pysteps_amd.extrapolation import get_method
    from tools import synth

    m, n = 96, 128
    members = [synth.rain_field_db(m, n, seed=s) for s in range(3)]
    vel = synth.true_velocity(m, n)
    got = parallel.sharded_extrapolate(members, vel, 2, rank=0, world_size=1, outval=-15.0)
    assert sorted(got) == [0, 1, 2]
    want = get_method("semilagrangian")(members[1], vel, 2, outval=-15.0)
    assert np.array_equal(got[1], want)


@pytest.mark.parametrize("world", [3, 8])
def test_row_band_tiling_equals_single_gpu(world):
    """Config-5 style output tiling with 'virtual ranks' on one GPU: the bands concatenate to the
    full-image result bit for bit (no halo exchange needed: inputs are replicated)."""
    from pysteps_amd import _lib, parallel
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.extrapolation import get_method
    from tools import synth

    m, n = 515, 384
    p = synth.rain_field_db(m, n, seed=8)
    y, x = np.mgrid[0:m, 0:n]
    v = synth.true_velocity(m, n) + np.stack([0.02 * (x - n / 2), -0.02 * (y - m / 2)]).astype(np.float32)
    dp, dv = DeviceArray.from_host(p), DeviceArray.from_host(v)
    full = get_method("semilagrangian")(dp, dv, 5, n_iter=2).to_host()
    for variant in (0, 7):  # the default selection (window kernel) and the gather kernels on row bands
        _lib.check(_lib.lib().psh_set_option(b"semilag_variant", variant))
        try:
            bands = []
            for rank in range(world):
                rows, out = parallel.tiled_extrapolate(dp, dv, 5, rank, world, n_iter=2)
                assert out.shape == (5, len(rows), n)
                bands.append(out.to_host())
            tiled = np.concatenate(bands, axis=1)
        finally:
            _lib.check(_lib.lib().psh_set_option(b"semilag_variant", 0))
        assert tiled.shape == full.shape
        assert np.array_equal(tiled, full, equal_nan=True)


def test_bench_multi_rank_path_runs_at_world_size_one():
    """bench.py's N > 1 leg (BASELINE config 4: RCCL communicator, ONE broadcast of [precip|u|v],
    member shard with perturbators from the seed chain, batched single-step calls) end to end on
    the one GPU a test box has."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--force-members-path", "--size", "512", "--leadtimes", "4",
         "--steps", "2", "--warmup", "1", "--members-per-gpu", "3"],
        cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = proc.stdout.strip().splitlines()
    assert len(lines) == 1, lines  # ONE JSON line on stdout, RCCL's banner included nowhere
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1
    assert "config 4" in line["config"]["workload"] and line["roofline"]["kernel"] == "semilag_members"
    assert line["value"] > 0 and line["config"]["broadcast_bytes"] == 3 * 512 * 512 * 4
    # the member loop of nowcasts.steps is what runs (update + advection), with the in-run base beside it
    assert "LK+semilag" not in line["metric"] and line["config"]["member_loop"]["updates_per_nowcast"] == 5
    assert line["single_gpu_base"]["value"] > 0
    assert 0.5 < line["weak_scaling_efficiency"] < 1.5  # world size 1: the same workload twice


def test_bench_advection_only_surrogate_still_runs():
    import json
    import subprocess
    import sys

    from conftest import ROOT

    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--force-members-path", "--advection-only", "--size", "512",
         "--leadtimes", "4", "--steps", "2", "--warmup", "1", "--members-per-gpu", "3"],
        cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert "advection only" in line["metric"] and line["config"]["member_loop"] is None and line["value"] > 0


def test_bench_config5_path_runs_at_world_size_one():
    """bench.py --workload config5 (row bands: banded LK through the RCCL allreduce / allgather calls,
    tiled semilag) on the one GPU a test box has, small grid"""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config5", "--size", "1024", "--leadtimes", "4",
         "--steps", "2", "--warmup", "1"],
        cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = proc.stdout.strip().splitlines()
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["scaling"] == "strong" and line["config"]["rccl_ranks"] == 1 and "config 5" in line["config"]["workload"]
    assert line["value"] > 0 and line["single_gpu_base"]["value"] > 0
    # at this size the step is the launch chain of the estimate, which the row bands make longer
    assert 0.05 < line["strong_scaling_efficiency"] < 1.5
