"""Wall clock of the REAL pysteps.nowcasts.steps (oracle/_ref) with the stock operators and with every
device piece this library offers for its member loop switched on (development aid / DESIGN.md 9).

    python tools/steps_quick.py [size] [members] [timesteps]
"""
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

from pysteps_amd import register  # noqa: E402
from tools import synth  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
members = int(sys.argv[2]) if len(sys.argv) > 2 else 4
timesteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
frames = synth.steps_frames(size, size, 3)
V = synth.true_velocity(size, size).astype(np.float64)
kw = dict(n_ens_members=members, n_cascade_levels=6, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, seed=42,
          vel_pert_method="bps", mask_method="incremental", probmatching_method="cdf", num_workers=1)
steps = nowcasts.get_method("steps")


def run(**extra):
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out = steps(frames, V, timesteps, **kw, **extra)
    return out, time.perf_counter() - t


want, stock_s = run(extrap_method="semilagrangian")
register.register(patch_main_loop=True, probmatching=True, autoregression=True, dilated_mask=True)
hip = dict(extrap_method="semilagrangian_hip", fft_method="hip", decomp_method="fft_hip", noise_method="nonparametric_hip")
run(**hip)  # first call: library initialisation, weight uploads
if os.environ.get("STEPS_PROFILE"):  # where the host time of the device run goes (cProfile, top of the list)
    import cProfile
    import pstats

    prof = cProfile.Profile()
    prof.enable()
    got, hip_s = run(**hip)
    prof.disable()
    stats = pstats.Stats(prof, stream=sys.stderr)
    stats.sort_stats("tottime").print_stats(28)
else:
    got, hip_s = run(**hip)
ok = np.isfinite(want) & np.isfinite(got)
print(json.dumps({"shape": [size, size], "members": members, "timesteps": timesteps, "stock_s": stock_s, "device_s": hip_s,
                  "speedup": stock_s / hip_s, "nan_masks_equal": bool(np.array_equal(np.isnan(want), np.isnan(got))),
                  "median_abs_diff": float(np.median(np.abs(want[ok] - got[ok])))}))
