#!/usr/bin/env python
"""bench.py - advection hot path (dense LK + semi-Lagrangian) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input that is
already resident in HBM: motion estimate from the input frames (dense
Lucas-Kanade) followed by the semi-Lagrangian extrapolation of the last frame
over T lead times.  Workload at N=1: BASELINE.json configs[2] (4096x4096 fp32,
2 input frames, 24 lead times, n_iter=1) - the configuration the metric is quoted
on.  With N>1 ranks (one per GPU, launched by torch.distributed.run) every rank
advects its own field ("members shard embarrassingly"): rank 0 synthesises the
inputs, one RCCL broadcast over xGMI distributes them before the timed region,
and there is no data-path collective afterwards (weak scaling).

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


def compulsory_bytes(m, n, T, K):
    """What one launch cannot avoid moving to / from HBM: the T output planes once, the three
    input planes once."""
    return (T + 3.0) * m * n * 4.0


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

    # ---- inputs: synthesised on rank 0, broadcast, resident in HBM ------------
    if dist.rank == 0:
        frames_d, vel_d = make_inputs(m, n, args.frames)
    else:
        frames_d = DeviceArray((args.frames, m, n), np.float32)
        vel_d = DeviceArray((2, m, n), np.float32)
    bcast_note = None
    if dist.world > 1:
        from pysteps_amd import parallel

        t0 = time.perf_counter()
        try:
            comm = parallel.Communicator(dist.rank, dist.world, dist.broadcast_bytes)
            comm.broadcast(frames_d, root=0)
            comm.broadcast(vel_d, root=0)
            synchronize()
            ok = 1.0
        except Exception as exc:  # RCCL unavailable: every rank synthesises the same seeded inputs
            bcast_note = "RCCL broadcast failed (%s); inputs synthesised per rank" % (exc,)
            ok = 0.0
        if dist.max(1.0 - ok) > 0.0:  # any rank failed -> all ranks fall back consistently
            if dist.rank != 0 or ok == 0.0:
                frames_d, vel_d = make_inputs(m, n, args.frames)
            bcast_note = bcast_note or "RCCL broadcast failed on another rank; inputs synthesised per rank"
        bcast_s = time.perf_counter() - t0
    else:
        bcast_s = None
    precip_d = frames_d.view(args.frames - 1)

    ev = [(Event(), Event()) for _ in range(args.steps)]

    def step(i=None):
        v = dense_lk(frames_d) if have_lk else vel_d
        if i is not None:
            ev[i][0].record()
        out = extrapolate(precip_d, v, T, outval=-15.0, n_iter=K)
        if i is not None:
            ev[i][1].record()
        return out

    for _ in range(args.warmup):
        step()
    synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = dist.max(elapsed)

    sl_ms = sum(a.elapsed_ms(b) for a, b in ev) / args.steps
    ms_per_step = elapsed / args.steps * 1e3
    value = dist.world * m * n * T / (ms_per_step * 1e-3) / 1e6
    b_alg = (16 * K + 8) if K > 0 else 16
    alg_bytes = float(b_alg) * m * n * T
    achieved = alg_bytes / (sl_ms * 1e-3) / 1e9
    workload = "%dx%d fp32, %d input frames, %s + semilag %d leadtimes n_iter=%d" % (
        m, n, args.frames, "dense LK" if have_lk else "true velocity (LK not timed)", T, K)

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
                "workload": workload,
                "lk_in_step": have_lk,
                "sharding": "one field per rank, inputs RCCL-broadcast before the timed region"
                if dist.world > 1 else "single GPU",
                "broadcast_s": bcast_s,
                "broadcast_note": bcast_note,
                # the two legs of the step (rank 0): the extrapolation kernel by HIP events,
                # the motion estimate (kernels + its two host hand-overs) as the rest of the step
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
        if dist.world == 1 and not args.no_host_path:
            line["config"].update(host_path(frames_d, vel_d, T, K))
        if not args.no_cpu_baseline and dist.world == 1:
            line["cpu_baseline"] = cpu_baseline(frames_d, vel_d, K, args.cpu_sample_steps, T, have_lk)
        print(json.dumps(line))
    dist.close()


if __name__ == "__main__":
    main()
