#!/usr/bin/env python
"""bench.py - advection hot path (dense LK + semi-Lagrangian) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input that is
already resident in HBM: motion estimate from the input frames (dense
Lucas-Kanade) followed by the semi-Lagrangian extrapolation of the last frame
over T lead times.  Workload at N=1: BASELINE.json configs[2] (4096x4096 fp32,
2 input frames, 24 lead times, n_iter=1) - the configuration the metric is quoted
on.  With N>1 ranks (one per GPU, launched by torch.distributed.run) the workload is
BASELINE.json configs[3]: a STEPS ensemble with 6 members per GPU (48 on 8 GPUs);
rank 0 synthesises the inputs and estimates the motion, ONE RCCL broadcast over
xGMI distributes [precip | u | v] before the timed region, every rank advects its
own perturbed members and there is no data-path collective afterwards (weak
scaling; an RCCL failure makes the run exit non-zero).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the fields).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=4096, help="grid is size x size")
    ap.add_argument("--leadtimes", type=int, default=24)
    ap.add_argument("--n-iter", type=int, default=1)
    ap.add_argument("--frames", type=int, default=2, help="LK input frames")
    ap.add_argument("--no-lk", action="store_true", help="time the extrapolator only (true velocity)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the NumPy-in / NumPy-out leg")
    ap.add_argument("--no-spectral", action="store_true", help="skip the FFT / cascade decomposition leg")
    ap.add_argument("--force-members-path", action="store_true",
                    help="run the N > 1 code path (RCCL communicator, broadcast, member shard) at any world size")
    ap.add_argument("--no-members-leg", action="store_true", help="skip the config-4 single-GPU reference leg")
    ap.add_argument("--members-per-gpu", type=int, default=6, help="STEPS members per GPU (config 4: 48 on 8 GPUs)")
    ap.add_argument("--cpu-sample-steps", type=int, default=2)
    return ap.parse_args()


class Dist:
    """Control plane of the N-rank run (rendezvous, barrier, max-reduce) over torch.distributed."""

    def __init__(self, want):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.td = None
        if self.world > 1:
            import torch
            import torch.distributed as td

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            td.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.td, self.torch = td, torch
        if want != self.world:
            if self.rank == 0:
                print("warning: --gpus %d but WORLD_SIZE=%d; using %d" % (want, self.world, self.world),
                      file=sys.stderr)

    def barrier(self):
        if self.td is not None:
            self.td.barrier()

    def max(self, value):
        if self.td is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])

    def broadcast_bytes(self, payload):
        """Small host-side broadcast (RCCL unique id)."""
        if self.td is None:
            return payload
        box = [payload]
        self.td.broadcast_object_list(box, src=0)
        return box[0]

    def close(self):
        if self.td is not None:
            self.td.destroy_process_group()


def make_inputs(m, n, frames):
    """Seeded synthetic inputs: dB rain field + smooth true motion.

    The LK input frames are the base field advected 1, 2, ... steps by the true
    motion; they are produced on the GPU by the extrapolator itself (device
    resident, outside the timed region)."""
    from pysteps_amd import _lib, extrapolation
    from pysteps_amd.device import DeviceArray
    from tools import synth

    base = synth.rain_field_db(m, n)
    vel = synth.true_velocity(m, n)
    vel_d = DeviceArray.from_host(vel)
    frames_d = DeviceArray((frames, m, n), np.float32)
    lib = _lib.lib()
    _lib.check(lib.psh_memcpy_h2d(frames_d.ptr, base.ctypes.data, base.nbytes))
    if frames > 1:
        adv = extrapolation.get_method("semilagrangian")(frames_d.view(0), vel_d, frames - 1, outval=-15.0)
        _lib.check(lib.psh_memcpy_d2d(frames_d.view(1).ptr, adv.ptr, adv.nbytes))
    _lib.check(lib.psh_sync())
    return frames_d, vel_d


def cpu_baseline(frames_d, vel_d, n_iter, sample_steps, leadtimes, with_lk):
    """CPU baseline on a BOUNDED sample of the same workload, timed on this box's host cores.

    Semi-Lagrangian leg: the restated driver over scipy.ndimage.map_coordinates (the
    third-party kernel the reference itself executes; single-threaded like the
    reference) on the full grid for `sample_steps` lead steps; cost is linear in the
    number of lead steps.  LK leg (if the step includes LK): the NumPy restatement
    of the OpenCV front end plus the cKDTree IDW (oracle/lk_opencv.py) on the
    top-left quarter-by-quarter crop (1/16 of the pixels, same feature budget);
    only its pixel-proportional part (everything after the sparse tracker) is
    scaled by 16, the sparse part is left unscaled (conservative)."""
    from oracle import lk_opencv as olk
    from oracle import semilag as osl
    from oracle import semilag_cport as ocl

    vel_h = vel_d.to_host()
    frames_h = frames_d.to_host()
    m, n = vel_h.shape[1:]
    # the UNMODIFIED reference function where oracle/_ref travelled with the snapshot
    # (oracle/build_ref.py); otherwise the restated driver over the same SciPy kernel
    ref_extrapolate = None
    try:
        from oracle import build_ref

        if build_ref.available():
            build_ref.activate()
            from pysteps.extrapolation.semilagrangian import extrapolate as ref_extrapolate
    except Exception:
        ref_extrapolate = None
    t0 = time.perf_counter()
    if ref_extrapolate is not None:
        ref_extrapolate(frames_h[-1], vel_h, sample_steps, outval=-15.0, n_iter=n_iter)
    else:
        osl.extrapolate(frames_h[-1], vel_h, sample_steps, outval=-15.0, n_iter=n_iter, backend="scipy")
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    ocl.extrapolate(frames_h[-1], vel_h, sample_steps, outval=-15.0, n_iter=n_iter)
    dtc = time.perf_counter() - t0
    t_sl_full = dt / sample_steps * leadtimes
    sample = "semi-Lagrangian: %dx%d, %d of %d lead steps, n_iter=%d, %s, %.1f s" % (
        m, n, sample_steps, leadtimes, n_iter,
        "pysteps.extrapolation.semilagrangian.extrapolate (oracle/_ref, unmodified reference)"
        if ref_extrapolate is not None else "restated driver over scipy map_coordinates", dt)
    # member-parallel figure: the reference is single-threaded, a host would run one
    # process per core - P processes advect one lead step each on the full grid
    multi = None
    if ref_extrapolate is not None:
        import multiprocessing as mp

        procs = min(8, len(os.sched_getaffinity(0)))
        try:
            ctx = mp.get_context("fork")
            t0 = time.perf_counter()
            workers = [ctx.Process(target=ref_extrapolate, args=(frames_h[-1], vel_h, 1),
                                   kwargs=dict(outval=-15.0, n_iter=n_iter)) for _ in range(procs)]
            for w in workers:
                w.start()
            for w in workers:
                w.join()
            dtp = time.perf_counter() - t0
            if all(w.exitcode == 0 for w in workers):
                multi = {"value": procs * m * n / dtp / 1e6, "cores": procs,
                         "note": "%d processes x 1 lead step each, reference function, %.1f s" % (procs, dtp)}
        except Exception:
            multi = None
    t_lk_full = 0.0
    if with_lk:
        cm, cn = max(m // 4, 64), max(n // 4, 64)
        crop = np.ascontiguousarray(frames_h[:, :cm, :cn])
        t0 = time.perf_counter()
        olk.sparse_lucaskanade(crop)
        t_sparse = time.perf_counter() - t0
        t0 = time.perf_counter()
        olk.dense_lucaskanade(crop)
        t_lk = time.perf_counter() - t0
        scale = (m * n) / float(cm * cn)
        t_lk_full = t_sparse + max(t_lk - t_sparse, 0.0) * scale
        sample += "; LK: %dx%d crop, %.1f s (sparse part %.1f s unscaled, rest x%.0f)" % (cn, cm, t_lk, t_sparse, scale)
    return {
        "value": m * n * leadtimes / (t_sl_full + t_lk_full) / 1e6,
        "unit": "Mpx*leadsteps/s",
        "cores": 1,
        # semi-Lagrangian leg: the reference itself when oracle/_ref is present; LK leg: always the
        # restatement (OpenCV is absent from this image and from the GPU box, profiles/r02/a_cv2_probe.txt)
        "kind": "reference" if ref_extrapolate is not None else "port",
        "sample": sample,
        "semilag_only": {"value": m * n * sample_steps / dt / 1e6, "cores": 1},
        "semilag_reference_multiprocess": multi,
        "semilag_port_c_openmp": {"value": m * n * sample_steps / dtc / 1e6, "cores": ocl.num_threads(),
                                  "note": "oracle/semilag_c.c, fused per-pixel float64 port"},
    }


def host_path(frames_d, vel_d, T, K, reps=3):
    """The plugin boundary itself: NumPy arrays in, NumPy arrays out through
    extrapolation.get_method("semilagrangian") (psh_semilag_host) - PCIe included.  Never `value`."""
    from pysteps_amd import extrapolation

    ex = extrapolation.get_method("semilagrangian")
    p, v = frames_d.view(frames_d.shape[0] - 1).to_host(), vel_d.to_host()
    m, n = p.shape
    ex(p, v, T, outval=-15.0, n_iter=K)  # warm-up: pins the result block, fills the block caches
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = ex(p, v, T, outval=-15.0, n_iter=K)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        del out
    moved = (3 + T) * m * n * 4.0
    return {"host_path_ms": best * 1e3, "host_path_pcie_gbs": moved / best / 1e9,
            "host_path_note": "numpy in / numpy out, %d x %d x %d lead times, best of %d; %.2f GB over PCIe" % (
                m, n, T, reps, moved / 1e9)}


def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary, if it matches."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
        return table.get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 vector peak
LK_STREAM_KERNELS = ("lk_stats1", "lk_stats1_final", "lk_open_vec", "lk_open", "lk_open_final", "lk_to_u8",
                     "lk_corner_response", "lk_max_final", "lk_corner_select", "lk_pyrdown", "lk_scharr")
LK_STREAM_BYTES_PER_PX_PAIR = 42.5  # ~40-45 B / pixel / frame pair of compulsory streaming


def roofline_lk(frames_d, m, n, pairs):
    """Second-tier rooflines of the motion estimate, from the committed rocprofv3 kernel
    averages of the SAME workload (profiles/kernel_stats_latest.json, written by
    tools/collect_profiles.py): the IDW kernel against the FP32 vector peak with the brute-force
    work model L x 8 flop per pixel, the streaming image passes against HBM with ~42.5 B per pixel
    and frame pair.  None if the committed profile is for another size."""
    path = os.path.join(ROOT, "profiles", "kernel_stats_latest.json")
    try:
        with open(path) as fh:
            prof = json.load(fh)
        if prof.get("workload") not in (None, "%dx%d" % (m, n)):
            return None
        kern = prof["kernels"]
    except Exception:
        return None
    from pysteps_amd import motion
    from pysteps_amd.utils import decluster

    xy, uv = motion.get_method("LK")(frames_d.to_host(), dense=False)
    L = len(decluster(xy, uv, 20, 1)[0]) if len(xy) else 0
    out = {"source": prof.get("source"), "vectors_interpolated": L}
    if "idw_fine" in kern and L:
        ms = (kern["idw_fine"]["avg_ns"] + kern.get("idw_coarse", {}).get("avg_ns", 0.0)) / 1e6
        flops = 8.0 * L * m * n
        out["idw"] = {"kernel": "idw_coarse + idw_fine", "bound": "fp32 valu", "kernel_ms": ms,
                      "model_flops": flops, "achieved": flops / (ms * 1e-3) / 1e12, "peak": FP32_VALU_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
                      "note": "work model = brute force over all L vectors; the kernel prunes to ~60 candidates per "
                              "tile, so frac > what the VALUs really execute"}
    ns = sum(kern[k]["ns_per_step"] for k in LK_STREAM_KERNELS if k in kern)
    if ns:
        nbytes = LK_STREAM_BYTES_PER_PX_PAIR * m * n * pairs
        out["image_passes"] = {"kernels": [k for k in LK_STREAM_KERNELS if k in kern], "bound": "hbm",
                               "ms_per_step": ns / 1e6, "alg_bytes_per_step": nbytes,
                               "achieved": nbytes / (ns * 1e-9) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": nbytes / (ns * 1e-9) / 1e9 / HBM_PEAK_GBS,
                               "per_kernel_ms": {k: kern[k]["ns_per_step"] / 1e6 for k in LK_STREAM_KERNELS if k in kern}}
    return out


def compulsory_bytes(m, n, T, K):
    """What one launch cannot avoid moving to / from HBM: the T output planes once, the three
    input planes once."""
    return (T + 3.0) * m * n * 4.0


class stdout_to_stderr:
    """RCCL prints its version banner with printf on stdout (buffered by libc until exit): route the
    file descriptor to stderr while the communicator comes up, so that stdout carries the ONE JSON
    line and nothing else."""

    def __enter__(self):
        import ctypes

        self.libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def members_workload(precip_d, vel_d, n_members, first_member, n_total, T, K):
    """BASELINE config 4 on one rank: `n_members` of the `n_total` STEPS members (global indices
    first_member ...), each with its own BPS-perturbed motion (perturbators recomputed from the
    ensemble seed, pysteps/nowcasts/steps.py:885-933), advected through T single-step stateful calls
    of the member-batched kernel - the extrapolation share of nowcasts.steps' main loop
    (pysteps/nowcasts/utils.py:441-462).  Returns step(): one full T-step nowcast of the members."""
    from pysteps_amd import _lib
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector, steps_perturbators

    m, n = precip_d.shape
    timestep_min = 5.0
    perts = steps_perturbators(n_total, 42, 1.0, timestep_min)[first_member:first_member + n_members]
    members_d = DeviceArray((n_members, m, n), np.float32)
    for j in range(n_members):
        _lib.check(_lib.lib().psh_memcpy_d2d(members_d.view(j).ptr, precip_d.ptr, precip_d.nbytes), "d2d")
    adv = EnsembleAdvector(vel_d, n_members, perts, n_iter=K, outval=-15.0)

    def step():
        adv.reset()
        out = None
        for t in range(T):
            out = adv.step(members_d, 1.0, timestep_min * (t + 1))
        return out

    return step


def spectral_leg(m, n):
    """SURVEY 8f rank 3 (first pieces): what the STEPS member loop calls per member and lead time -
    the FFT method object, the cascade decomposition and the CDF matching - resident on the device (HIP
    events) with numpy's pocketfft timed once beside it.  Reported under config, not part of `value`."""
    from pysteps_amd.device import DeviceArray, Event, synchronize
    from pysteps_amd.utils.fft import get_hip, supported_shape

    if not supported_shape((m, n)):
        return {}
    rng = np.random.default_rng(7)
    x = rng.standard_normal((m, n))
    fft = get_hip((m, n))
    dx = DeviceArray.from_host(x)
    dX = fft.rfft2(dx)
    fft.irfft2(dX)
    synchronize()
    reps = 5
    e0, e1, e2 = Event(), Event(), Event()
    e0.record()
    for _ in range(reps):
        dX = fft.rfft2(dx)
    e1.record()
    for _ in range(reps):
        fft.irfft2(dX)
    e2.record()
    synchronize()
    fwd, inv = e0.elapsed_ms(e1) / reps, e1.elapsed_ms(e2) / reps
    t = time.perf_counter()
    np.fft.rfft2(x)
    cpu = (time.perf_counter() - t) * 1e3
    # compulsory traffic of the two-pass transform: real plane in, half spectrum out, in and out again
    spec = m * (n // 2 + 1) * 16
    traffic = m * n * 8 + 3 * spec
    out = {"rfft2_ms": fwd, "irfft2_ms": inv, "rfft2_hbm_frac": traffic / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "numpy_rfft2_ms": cpu, "dtype": "f64"}
    try:
        from pysteps_amd.cascade import decomposition_fft

        nlev = 8
        # Gaussian-shaped band weights of the reference's form (levels, m, n/2+1); their values do not
        # change the work
        ky, kx = np.fft.fftfreq(m)[:, None], np.fft.rfftfreq(n)[None, :]
        r = np.hypot(ky, kx)
        centres = 0.5 * 2.0 ** -np.arange(nlev)[::-1]
        w = np.stack([np.exp(-0.5 * ((np.log2(r + 1e-9) - np.log2(c)) / 0.6) ** 2) for c in centres])
        bp = {"weights_2d": w / w.sum(axis=0, keepdims=True), "weights_1d": np.zeros((nlev, 1)), "shape": (m, n)}
        decomposition_fft(dx, bp, normalize=True, compute_stats=True)  # uploads and caches the weights
        synchronize()
        e3, e4 = Event(), Event()
        e3.record()
        decomposition_fft(dx, bp, normalize=True, compute_stats=True)
        e4.record()
        synchronize()
        out["cascade_decompose_%d_levels_ms" % nlev] = e3.elapsed_ms(e4)
    except Exception as exc:  # the spectral leg never takes the headline down
        out["cascade_note"] = "%s: %s" % (type(exc).__name__, exc)
    try:
        from pysteps_amd.postprocessing.probmatching import nonparam_match_empirical_cdf

        # a continuous forecast with 25 % of its pixels above the zero value against an observation
        # quantised to 0.1 dB with 35 % wet pixels (the call of nowcasts/steps.py:1199)
        y = rng.standard_normal((m, n))
        d_fct = DeviceArray.from_host(np.where(x > 0.6745, 5.0 * x, -15.0))
        d_obs = DeviceArray.from_host(np.round(np.where(y > 0.3853, 5.0 * y, -15.0), 1))
        nonparam_match_empirical_cdf(d_fct, d_obs)
        synchronize()
        e5, e6 = Event(), Event()
        e5.record()
        for _ in range(reps):
            nonparam_match_empirical_cdf(d_fct, d_obs)
        e6.record()
        synchronize()
        out["probmatch_cdf_ms"] = e5.elapsed_ms(e6) / reps
    except Exception as exc:
        out["probmatch_note"] = "%s: %s" % (type(exc).__name__, exc)
    return out


def time_steps(step, dist, steps, warmup, events=None):
    from pysteps_amd.device import synchronize

    for _ in range(warmup):
        step()
    synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        if events is not None:
            events[i][0].record()
        step()
        if events is not None:
            events[i][1].record()
    synchronize()
    dist.barrier()
    return dist.max(time.perf_counter() - t0)


def main():
    args = parse_args()
    dist = Dist(args.gpus)
    os.environ.setdefault("PYSTEPS_HIP_DEVICE", str(dist.local_rank))

    from pysteps_amd.device import DeviceArray, Event, synchronize
    from pysteps_amd import extrapolation

    m = n = args.size
    T, K = args.leadtimes, args.n_iter
    extrapolate = extrapolation.get_method("semilagrangian")

    have_lk = False
    dense_lk = None
    if not args.no_lk:
        try:
            from pysteps_amd import motion

            dense_lk = motion.get_method("LK")
            have_lk = True
        except (ImportError, AttributeError):
            have_lk = False

    if dist.world > 1 or args.force_members_path:
        return main_members(args, dist, dense_lk if have_lk else None)

    # ---- N = 1: BASELINE configs[2], inputs synthesised and resident in HBM -------------
    frames_d, vel_d = make_inputs(m, n, args.frames)
    precip_d = frames_d.view(args.frames - 1)

    ev = [(Event(), Event()) for _ in range(args.steps)]
    counter = {"i": None}

    def step():
        v = dense_lk(frames_d) if have_lk else vel_d
        i = counter["i"]
        if i is not None:
            ev[i][0].record()
        out = extrapolate(precip_d, v, T, outval=-15.0, n_iter=K)
        if i is not None:
            ev[i][1].record()
            counter["i"] = i + 1
        return out

    for _ in range(args.warmup):
        step()
    synchronize()
    dist.barrier()
    counter["i"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    synchronize()
    dist.barrier()
    elapsed = dist.max(time.perf_counter() - t0)
    counter["i"] = None

    sl_ms = sum(a.elapsed_ms(b) for a, b in ev) / args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = m * n * T / (ms_per_step * 1e-3) / 1e6
    b_alg = (16 * K + 8) if K > 0 else 16
    alg_bytes = float(b_alg) * m * n * T
    achieved = alg_bytes / (sl_ms * 1e-3) / 1e9
    workload = "%dx%d fp32, %d input frames, %s + semilag %d leadtimes n_iter=%d" % (
        m, n, args.frames, "dense LK" if have_lk else "true velocity (LK not timed)", T, K)
    line = {
        "metric": "Mpixels*leadsteps/s (LK+semilag) at %dx%d fp32" % (m, n),
        "value": value,
        "unit": "Mpx*leadsteps/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "lk_in_step": have_lk,
            "sharding": "single GPU",
            # the two legs of the step: the extrapolation (velocity packing + kernel) by
            # HIP events, the motion estimate as the rest of the step
            "semilag_only_mpx_leadsteps_s": m * n * T / (sl_ms * 1e-3) / 1e6,
            "lk_ms_per_step": (ms_per_step - sl_ms) if have_lk else None,
        },
        "roofline": {
            "kernel": "semilag_fused",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "alg_bytes_per_launch": alg_bytes,
            "kernel_ms": sl_ms,
            "traffic": pmc_traffic("semilag_%dx%d_T%d_K%d" % (m, n, T, K)),
        },
    }
    # the contract's `frac` prices ALGORITHMIC bytes; the inputs are served from L2/MALL, so the
    # DRAM-side picture is given beside it: counter traffic over the same duration, and the time
    # the compulsory stream alone would need at peak
    rf = line["roofline"]
    rf["hbm_frac"] = (rf["traffic"] / (sl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rf["traffic"] else None
    rf["floor_ms"] = compulsory_bytes(m, n, T, K) / (HBM_PEAK_GBS * 1e9) * 1e3
    if have_lk:
        line["roofline_lk"] = roofline_lk(frames_d, m, n, args.frames - 1)
    if not args.no_host_path:
        line["config"].update(host_path(frames_d, vel_d, T, K))
    if not args.no_spectral:
        line["config"]["spectral"] = spectral_leg(m, n)
    if not args.no_members_leg:
        # what ONE GPU of the N > 1 runs does (config 4: members_per_gpu members, T single-step
        # stateful calls): the single-GPU figure the multi-GPU values are to be compared with
        mstep = members_workload(precip_d, vel_d, args.members_per_gpu, 0, args.members_per_gpu, T, K)
        el = time_steps(mstep, dist, 3, 1)
        line["config"]["config4_one_gpu"] = {
            "members": args.members_per_gpu,
            "ms_per_step": el / 3 * 1e3,
            "value": args.members_per_gpu * m * n * T / (el / 3) / 1e6,
            "note": "same per-GPU workload as the --gpus N > 1 runs (weak scaling reference)",
        }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(frames_d, vel_d, K, args.cpu_sample_steps, T, have_lk)
    print(json.dumps(line))
    dist.close()


def main_members(args, dist, dense_lk):
    """N > 1 (one rank per GPU): BASELINE config 4, `members_per_gpu` STEPS members per GPU.

    Rank 0 synthesises the frames, estimates the motion field and packs [precip | u | v] into one
    buffer; ONE RCCL broadcast over xGMI (192 MiB at 4096^2) hands it to every rank before the
    timed region; each rank then advects its own members (partition of 6N members, perturbators
    recomputed from the ensemble seed) with no data-path collective (weak scaling).  An RCCL failure
    is fatal: the run exits non-zero instead of reporting numbers of a run that did not communicate."""
    from pysteps_amd import _lib, parallel
    from pysteps_amd.device import DeviceArray, Event, synchronize

    m = n = args.size
    T, K = args.leadtimes, args.n_iter
    pack = DeviceArray((3, m, n), np.float32)
    if dist.rank == 0:
        frames_d, vel_d = make_inputs(m, n, args.frames)
        v = dense_lk(frames_d) if dense_lk is not None else vel_d
        lib = _lib.lib()
        _lib.check(lib.psh_memcpy_d2d(pack.view(0).ptr, frames_d.view(args.frames - 1).ptr, m * n * 4), "d2d")
        _lib.check(lib.psh_memcpy_d2d(pack.view(1).ptr, v.ptr, 2 * m * n * 4), "d2d")
        synchronize()
        del frames_d, vel_d, v
    t0 = time.perf_counter()
    ok = 1.0
    err = None
    try:
        with stdout_to_stderr():
            comm = parallel.Communicator(dist.rank, dist.world, dist.broadcast_bytes)
            comm.broadcast(pack, root=0)
            synchronize()
    except Exception as exc:
        ok, err = 0.0, exc
    if dist.max(1.0 - ok) > 0.0:
        print("rank %d: RCCL broadcast failed on %s: %s" % (
            dist.rank, "this rank" if ok == 0.0 else "another rank", err), file=sys.stderr)
        dist.close()
        sys.exit(3)
    bcast_s = time.perf_counter() - t0
    precip_d = pack.view(0)
    vel_d = DeviceArray((2, m, n), np.float32, ptr=pack.view(1).ptr, owner=pack)

    per = args.members_per_gpu
    n_total = per * dist.world
    mine = parallel.partition(n_total, dist.world, dist.rank)
    step = members_workload(precip_d, vel_d, len(mine), mine.start, n_total, T, K)
    ev = [(Event(), Event()) for _ in range(args.steps)]
    elapsed = time_steps(step, dist, args.steps, args.warmup, ev)
    kernel_ms = sum(a.elapsed_ms(b) for a, b in ev) / args.steps / T  # one batched launch
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * m * n * T / (ms_per_step * 1e-3) / 1e6
    # stateful single-step call: D read + write 16, three velocity passes 8 each
    # (increment rebuild, midpoint, end point) for n_iter = 1, field 4, store 4 -> 48 B / px / member
    b_alg = (24 + 16 * K + 8) if K > 0 else 32
    alg_bytes = float(b_alg) * m * n * len(mine)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    if dist.rank == 0:
        line = {
            "metric": "Mpixels*leadsteps/s (LK+semilag) at %dx%d fp32" % (m, n),
            "value": value,
            "unit": "Mpx*leadsteps/s",
            "n_gpus": dist.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d fp32, %d-member STEPS ensemble advection (BASELINE config 4), %d members per GPU, "
                            "%d lead steps as single-step stateful calls with BPS velocity perturbations, n_iter=%d" % (
                                m, n, n_total, per, T, K),
                "sharding": "members partitioned over ranks, motion field from %s on rank 0, [precip|u|v] in ONE "
                            "RCCL broadcast before the timed region, no data-path collective" % (
                                "dense LK" if dense_lk is not None else "the synthetic truth"),
                "rccl_ranks": dist.world,
                "broadcast_s": bcast_s,
                "broadcast_bytes": pack.nbytes,
                "compare_with": "config.config4_one_gpu.value of the --gpus 1 line (same per-GPU workload)",
            },
            "roofline": {
                "kernel": "semilag_members",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "alg_bytes_per_launch": alg_bytes,
                "kernel_ms": kernel_ms,
                "traffic": None,
            },
        }
        print(json.dumps(line))
    dist.close()


if __name__ == "__main__":
    main()
