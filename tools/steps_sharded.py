"""BASELINE config 4 as the reference runs it: the REAL pysteps.nowcasts.steps (oracle/_ref), the ensemble
sharded over the ranks of a node - one process per GPU, ONE RCCL broadcast of the inputs, every rank runs
the nowcaster for ITS members (pysteps_amd.parallel.steps_shard: its slice of the ensemble's seed chain)
with the resident member loop, results stay sharded.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        tools/steps_sharded.py [size] [members] [timesteps]
    python tools/steps_sharded.py 256 6 3 --virtual-ranks 3     # one process: the shards one after the other,
                                                                # compared with the whole ensemble run at once

Prints one JSON line on rank 0 (seconds = the slowest rank).  Development aid / test driver - bench.py may not
drive oracle/_ref in its timed region; its N > 1 leg times the same member loop on a synthetic state.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import build_ref

build_ref.activate()
from pysteps import nowcasts  # noqa: E402

import bench  # noqa: E402
from pysteps_amd import parallel, register  # noqa: E402
from tools import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("size", nargs="?", type=int, default=1024)
ap.add_argument("members", nargs="?", type=int, default=6, help="members of the WHOLE ensemble")
ap.add_argument("timesteps", nargs="?", type=int, default=3)
ap.add_argument("--virtual-ranks", type=int, default=0)
ap.add_argument("--seed", type=int, default=42)
args = ap.parse_args()
m = n = args.size
KW = dict(n_cascade_levels=6, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, mask_method="incremental",
          probmatching_method="cdf", num_workers=1, extrap_method="semilagrangian_hip", fft_method="hip",
          decomp_method="fft_hip", noise_method="nonparametric_hip", vel_pert_method="bps_hip")


def run(frames, V, **shard_kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return nowcasts.get_method("steps")(frames, V, args.timesteps, **dict(KW, **shard_kw))


register.register(patch_main_loop=True, probmatching=True, autoregression=True, dilated_mask=True)
if args.virtual_ranks:
    frames = synth.steps_frames(m, n, 3).astype(np.float64)
    V = synth.true_velocity(m, n).astype(np.float64)
    whole = run(frames, V, n_ens_members=args.members, seed=args.seed)
    parts = []
    for r in range(args.virtual_ranks):
        members, kw = parallel.steps_shard(args.seed, args.members, args.virtual_ranks, r)
        if len(members):
            parts.append(run(frames, V, **kw))
    got = np.concatenate(parts, axis=0)
    print(json.dumps({"shape": [m, n], "members": args.members, "virtual_ranks": args.virtual_ranks,
                      "shards_equal_whole_ensemble": bool(np.array_equal(got, whole, equal_nan=True)),
                      "max_abs_diff": float(np.nanmax(np.abs(got - whole)))}))
    sys.exit(0)

dist = bench.Dist(int(os.environ.get("WORLD_SIZE", "1")))
os.environ.setdefault("PYSTEPS_HIP_DEVICE", str(dist.local_rank))
from pysteps_amd.device import DeviceArray, synchronize  # noqa: E402

pack = DeviceArray((5, m, n), np.float64)  # [3 frames | u | v]
if dist.rank == 0:
    host = np.concatenate([synth.steps_frames(m, n, 3).astype(np.float64), synth.true_velocity(m, n).astype(np.float64)])
    pack = DeviceArray.from_host(host)
with bench.stdout_to_stderr():
    comm = parallel.Communicator(dist.rank, dist.world, dist.broadcast_bytes)
    comm.broadcast(pack, root=0)
    synchronize()
host = pack.to_host()
frames, V = host[:3], host[3:]
members, kw = parallel.steps_shard(args.seed, args.members, dist.world, dist.rank)
run(frames, V, **dict(kw, n_ens_members=max(1, min(2, len(members)))))  # warm-up: library, weights, block pools
dist.barrier()
t0 = time.perf_counter()
out = run(frames, V, **kw) if len(members) else None
synchronize()
dist.barrier()
seconds = dist.max(time.perf_counter() - t0)
if dist.rank == 0:
    print(json.dumps({"shape": [m, n], "members": args.members, "timesteps": args.timesteps, "ranks": dist.world,
                      "members_of_rank0": list(members), "seconds": seconds,
                      "mpx_leadsteps_per_s": args.members * m * n * args.timesteps / seconds / 1e6,
                      "result_of_rank0": None if out is None else list(out.shape)}))
dist.close()
