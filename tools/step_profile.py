"""cProfile of one full LK+SL step with resident inputs (development aid)."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from pysteps_amd.device import synchronize
from pysteps_amd import extrapolation, motion

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames_d, vel_d = bench.make_inputs(m, n, 2)
lk = motion.get_method("LK"); ex = extrapolation.get_method("semilagrangian")
precip = frames_d.view(1)
def step():
    v = lk(frames_d)
    out = ex(precip, v, 24, outval=-15.0)
    synchronize()
for _ in range(3): step()
t0 = time.perf_counter()
for _ in range(5): step()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
