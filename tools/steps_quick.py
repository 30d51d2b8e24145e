"""Wall clock of the REAL pysteps.nowcasts.steps (oracle/_ref): the stock operators against every device
piece of this library - extrapolator, member-batched advection and the resident member update
(cascades, random streams, masks in HBM; pysteps_amd/nowcasts/steps_resident.py).

    python tools/steps_quick.py [size] [members] [timesteps] [--stock-members B] [--stock-steps T] [--no-stock]

At BASELINE config 4's size (4096^2, 6 members per GPU) the stock run takes minutes per member and
time step, so it is SAMPLED: ``--stock-members`` x ``--stock-steps`` (same seed: member j / step t of the
sample are member j / step t of the full run, which is also what the parity figures compare) and its
main-loop time is scaled by member-steps; initialisation is not scaled.
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

from pysteps_amd import register  # noqa: E402
from pysteps_amd.nowcasts import utils as hip_loop  # noqa: E402
from tools import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("size", nargs="?", type=int, default=1024)
ap.add_argument("members", nargs="?", type=int, default=4)
ap.add_argument("timesteps", nargs="?", type=int, default=3)
ap.add_argument("--stock-members", type=int, default=None)
ap.add_argument("--stock-steps", type=int, default=None)
ap.add_argument("--no-stock", action="store_true")
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--no-resident", action="store_true", help="round-2 state: device operators, host member loop")
ap.add_argument("--domain", default="spatial", choices=["spatial", "spectral"], help="nowcasts.steps(domain=...)")
ap.add_argument("--profile", action="store_true", help="cProfile of the device run (host side), top of the list to stderr")
args = ap.parse_args()
size, members, timesteps = args.size, args.members, args.timesteps
frames = synth.steps_frames(size, size, 3)
V = synth.true_velocity(size, size).astype(np.float64)
kw = dict(n_cascade_levels=args.levels, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, seed=42, vel_pert_method="bps",
          mask_method="incremental", probmatching_method="cdf", num_workers=1, measure_time=True, domain=args.domain)
steps = nowcasts.get_method("steps")


def run(n_members, n_steps, **extra):
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out, init_s, loop_s = steps(frames, V, n_steps, n_ens_members=n_members, **dict(kw, **extra))
    return out, dict(total_s=time.perf_counter() - t, init_s=init_s, loop_s=loop_s)


report = {"shape": [size, size], "members": members, "timesteps": timesteps, "cascade_levels": args.levels, "domain": args.domain}
want = None
if not args.no_stock:
    sm = args.stock_members or members
    st = args.stock_steps or timesteps
    want, stock = run(sm, st, extrap_method="semilagrangian")
    stock.update(members=sm, timesteps=st)
    # T + 1 member updates and T advections per member: scale the loop by member-steps
    stock["loop_s_scaled_to_full"] = stock["loop_s"] * (members * (timesteps + 1)) / (sm * (st + 1))
    report["stock"] = stock
register.register(patch_main_loop=True, probmatching=True, autoregression=True, dilated_mask=True)
hip_loop.resident_update_enabled = not args.no_resident
hip = dict(extrap_method="semilagrangian_hip", fft_method="hip", decomp_method="fft_hip", noise_method="nonparametric_hip",
           vel_pert_method="bps_hip")
run(members, timesteps if size <= 2048 else 1, **hip)  # library initialisation, weight uploads, block pools
if size > 2048:
    run(members, timesteps, **hip)  # ... and the pinned result block of this very shape (a service repeats its shape)
if args.profile:
    import cProfile
    import pstats

    prof = cProfile.Profile()
    prof.enable()
    got, dev = run(members, timesteps, **hip)
    prof.disable()
    pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(32)
else:
    got, dev = run(members, timesteps, **hip)
dev["resident"] = not args.no_resident
if hip_loop.last_run_stats and not args.no_resident:
    ph = dict(hip_loop.last_run_stats)
    dev["device_ms_by_phase"] = {k: round(v, 3) for k, v in ph.items() if isinstance(v, float)}
    moved = ph.get("upload", 0.0) + ph.get("download", 0.0) + ph.get("result_block", 0.0)
    busy = sum(v for v in ph.values() if isinstance(v, float))
    dev["transfer_share_of_loop_device_time"] = moved / busy if busy else None
    dev["ms_per_member_update"] = ph.get("update", 0.0) / (members * (timesteps + 1))
report["device"] = dev
if want is not None:
    sub = got[: want.shape[0], : want.shape[1]]
    ok = np.isfinite(want) & np.isfinite(sub)
    scale = float(np.nanmax(want) - np.nanmin(want))
    diff = np.abs(want[ok] - sub[ok])
    report["parity_on_the_stock_sample"] = {
        "nan_masks_equal": bool(np.array_equal(np.isnan(want), np.isnan(sub))),
        "median_abs_diff": float(np.median(diff)),
        "pixels_off_by_more_than_1e-2_of_range": float(np.count_nonzero(diff > 1e-2 * scale) / diff.size),
        "rel_l2_of_the_rest": float(np.linalg.norm(diff[diff <= 1e-2 * scale]) / np.linalg.norm(want[ok])),
    }
    report["speedup_loop"] = report["stock"]["loop_s_scaled_to_full"] / dev["loop_s"]
print(json.dumps(report))
