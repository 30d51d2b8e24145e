"""World-size-2 CPU worker (gloo) for tests/test_parallel_cpu.py - launched by torch.distributed.run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pysteps_amd import parallel  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist = bench.Dist(want=2)
    assert dist.world == 2
    # control plane used by bench.py: barrier, max over ranks, unique-id style broadcast
    dist.barrier()
    slowest = dist.max(1.0 + dist.rank)
    token = dist.broadcast_bytes(b"x" * 128 if dist.rank == 0 else None)
    # member sharding: 48 members, 2 ranks -> 24 each, disjoint and complete
    mine = list(parallel.partition(48, dist.world, dist.rank))
    owners = [parallel.owner_of(j, 48, dist.world) for j in mine]
    # config 4 in bench.py: every rank recomputes the perturbators of ITS members from the seed
    from pysteps_amd.extrapolation.ensemble import steps_noise_generators, steps_perturbators

    per_gpu = 6
    n_total = per_gpu * dist.world
    shard = parallel.partition(n_total, dist.world, dist.rank)
    eps = [p["eps_par"] for p in steps_perturbators(n_total, 42, 1.0, 5.0)[shard.start:shard.stop]]
    # ... and the random streams of its members (the first draw of each identifies the stream)
    first_draws = [float(rs.standard_normal()) for rs in steps_noise_generators(n_total, 42)[shard.start:shard.stop]]
    # config 5: row bands of the extrapolation, the motion estimate replicated (no collective in the step)
    plan = parallel.config5_plan(8192, dist.world, dist.rank)
    with open(os.path.join(out_dir, "rank%d.json" % dist.rank), "w") as fh:
        json.dump({"rank": dist.rank, "max": slowest, "token_len": len(token), "mine": mine, "config5": plan,
                   "owners": owners, "shard": list(shard), "eps_par": eps, "first_draws": first_draws}, fh)
    dist.barrier()
    dist.close()


if __name__ == "__main__":
    main()
