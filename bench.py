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
# VALU issue peak of the chip in wave64 instructions per second: 256 CUs x 4 SIMDs, one wave64 VALU instruction occupies
# its SIMD for 4 cycles (measured for fp32 / fp64 / packed / DPP, profiles/r04/a_valu_probe.txt), 2.4 GHz max clock
# (MI355X_MICROARCH.md; the chip sustains 2.1-2.2 GHz under this kernel, so a frac of ~0.9 is the practical ceiling)
N_SIMD = 1024
VALU_PEAK_GINST = N_SIMD * 2.4 / 4.0
# the sources whose kernels the committed counter files describe: tools/collect_profiles.py stores their git blob
# hashes beside the counters, and a line whose sources differ reports `traffic: null, traffic_stale: true` instead of
# counters of a kernel that is no longer the one being timed
SL_SOURCES = ("pysteps_amd/csrc/semilag.hip", "pysteps_amd/csrc/semilag_device.h", "pysteps_amd/csrc/common.h")
MU_SOURCES = ("pysteps_amd/csrc/steps_loop.hip", "pysteps_amd/csrc/fft.hip", "pysteps_amd/csrc/probmatch.hip",
              "pysteps_amd/csrc/mask.hip", "pysteps_amd/csrc/cascade.hip", "pysteps_amd/csrc/common.h")
LK_SOURCES = ("pysteps_amd/csrc/lk.hip", "pysteps_amd/csrc/lk_sparse.hip", "pysteps_amd/csrc/sparse_qc.hip",
              "pysteps_amd/csrc/idw.hip", "pysteps_amd/csrc/dense_lk.hip", "pysteps_amd/csrc/common.h")


def blob_hash(relpath):
    """git's blob hash of a file of the working tree (what `git hash-object` prints)."""
    import hashlib

    try:
        with open(os.path.join(ROOT, relpath), "rb") as fh:
            data = fh.read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def sources_match(recorded, sources):
    """True if the blob hashes stored with a counter file are those of the tree's sources."""
    return bool(recorded) and all(recorded.get(src) == blob_hash(src) for src in sources)


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
    ap.add_argument("--cpu-sample-steps", type=int, default=6)
    ap.add_argument("--workload", choices=("default", "config5"), default="default",
                    help="config5: 8192^2 x 36 lead times, row bands over the ranks (banded LK + tiled semilag, RCCL collectives)")
    ap.add_argument("--config5-lk", choices=("replicated", "banded"), default="replicated",
                    help="config5: the motion estimate on every rank (default: no collective in the step) or in row bands "
                         "(host-driven stage loop with 5 small collectives per estimate)")
    ap.add_argument("--no-steps-loop", action="store_true", help="skip the STEPS member-loop leg of the N = 1 line")
    ap.add_argument("--no-steps-stock", action="store_true", help="skip the sampled stock nowcasts.steps run (oracle/_ref) of steps_e2e")
    ap.add_argument("--advection-only", action="store_true",
                    help="N > 1: time the member advection alone (the SL-only surrogate of rounds 1-2) instead of the member loop")
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


def pmc_counters(workload):
    """(record, stale) of the extrapolation kernel from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json:
    HBM bytes per launch, SQ_INSTS_VALU, GRBM_GUI_ACTIVE, all per launch of the SAME workload); stale = the kernel's
    sources are not the ones the counters were taken from."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh).get(workload)
    except Exception:
        rec = None
    if not rec:
        return None, False
    return rec, not sources_match(rec.get("source_hashes"), SL_SOURCES)


# algorithmic bytes per pixel of the streaming image passes of the motion estimate (what each pass has
# to read and write once): float32 frame in / keep bits / uint8 renderings / float32 response
LK_PASS_BYTES = {
    "lk_stats1": (4.0, "frame read"),
    "lk_open_bits": (4.0 + 8.0 / 60.0, "frame read + one keep word per 60 pixels written (no cleaned frame)"),
    "lk_to_u8_bits": (4.0 + 8.0 / 60.0 + 1.5,
                      "frame + keep words read, tracking rendering written (+ feature rendering for the first frame of a pair)"),
    "lk_corner_response_cols": (5.0, "uint8 rendering read + float32 response written"),
    "lk_corner_select": (4.0, "response read"),
    "lk_pyrdown": (1.25 * 1.333, "per frame: every level read once, the next one written (geometric series)"),
}
LK_CALLS_PER_PAIR = {"lk_stats1": 2, "lk_open_bits": 2, "lk_to_u8_bits": 2, "lk_corner_response_cols": 1, "lk_corner_select": 1,
                     "lk_pyrdown": 2}


def roofline_lk(frames_d, m, n, pairs):
    """Second-tier rooflines of the motion estimate from the committed rocprofv3 summaries of
    the SAME workload (profiles/kernel_stats_latest.json, written by tools/collect_profiles.py from the
    kernel trace and the PMC passes): the streaming image passes against HBM, each with its own
    algorithmic bytes; the VALU-bound kernels (idw_fine3, lk_corner_response_cols) by the share of the
    chip's VALU issue slots their instructions take (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel
    cycles); 4 cycles per wave64 instruction measured, profiles/r04/a_valu_probe.txt) - a work model in
    flops would price what the kernel (rightly) does not execute."""
    path = os.path.join(ROOT, "profiles", "kernel_stats_latest.json")
    try:
        with open(path) as fh:
            prof = json.load(fh)
        if prof.get("workload") not in (None, "%dx%d" % (m, n)):
            return None
        kern = prof["kernels"]
    except Exception:
        return None
    counters = prof.get("counters", {})
    out = {"source": prof.get("source"), "counter_source": prof.get("counter_source"),
           "stale": not sources_match(prof.get("source_hashes"), LK_SOURCES),
           "stale_note": "true = the LK sources changed after these kernel times / counters were taken (tools/collect_profiles.py)"}
    passes = {}
    total_ns = total_bytes = 0.0
    for name, (bpp, what) in LK_PASS_BYTES.items():
        if name not in kern:
            continue
        calls = LK_CALLS_PER_PAIR[name] * pairs
        ns = kern[name]["ns_per_step"]
        nbytes = bpp * m * n * calls
        passes[name] = {"ms_per_step": ns / 1e6, "alg_bytes_per_px": bpp, "what": what,
                        "achieved": nbytes / (ns * 1e-9) / 1e9, "frac": nbytes / (ns * 1e-9) / 1e9 / HBM_PEAK_GBS}
        total_ns += ns
        total_bytes += nbytes
    if passes:
        out["image_passes"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "ms_per_step": total_ns / 1e6,
                               "alg_bytes_per_step": total_bytes, "achieved": total_bytes / (total_ns * 1e-9) / 1e9,
                               "frac": total_bytes / (total_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, "per_kernel": passes}
    valu = {}
    for name in ("idw_fine3", "idw_coarse", "lk_corner_response_cols", "lk_track_rows", "lk_open_bits", "lk_to_u8_bits",
                 "outliers_local_sorted"):
        c = counters.get(name)
        if not c or "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter adds the 8 XCDs up
        valu[name] = {"kernel_ms": kern.get(name, {}).get("avg_ns", 0.0) / 1e6, "valu_insts_per_launch": c["SQ_INSTS_VALU"],
                      "kernel_cycles": cycles, "valu_issue_frac": c["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * cycles)}
    if valu:
        out["valu_bound"] = {"bound": "valu issue", "definition": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); "
                             "a wave64 VALU instruction occupies its SIMD for 4 cycles whatever its type (fp32, fp64, packed, DPP), "
                             "transcendentals for 8 (profiles/r04/a_valu_probe.txt): a lower bound of the VALU busy share", "per_kernel": valu}
    return out


def semilag_roofline(m, n, T, K, sl_ms, alg_bytes):
    """The roofline block of the extrapolation leg.  `kernel_ms` is measured live (HIP events on the library stream
    around the ONE kernel of the leg).  The kernel that runs by default - the workgroup window - serves every tap from
    LDS, so the roof that binds it is VALU ISSUE, not HBM: `achieved` = wave64 VALU instructions per second
    (SQ_INSTS_VALU per launch, from the committed counter pass of this very workload, over the live kernel time),
    `peak` = 1024 SIMDs x 2.4 GHz / 4 cycles per instruction, `frac` their ratio.  `traffic` = HBM bytes per launch from
    the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction applied, tools/calib_copy.py).  The HBM reading the contract
     prescribes - ALGORITHMIC bytes / time / 8 TB/s - is kept under `hbm`: it prices every tap as a byte
    from memory, exceeds 1 for this kernel and says nothing about its distance from any bound; `hbm.hbm_frac`
    (counter bytes) and `hbm.kernel_over_floor` are the physical figures.  Counters are static per source revision:
    when the kernel's sources differ from the ones the counters were taken from, they are withheld."""
    from pysteps_amd import _lib

    choice = _lib.load().psh_semilag_kernel(m, n, T, K, 1, 1)
    kernel = {12: "semilag_window"}.get(choice, "semilag_fused (gather mode %d)" % choice)
    rec, stale = pmc_counters("semilag_%dx%d_T%d_K%d" % (m, n, T, K))
    if rec is not None and rec.get("kernel") != kernel.split(" ")[0]:
        rec, stale = None, False  # counters of another kernel
    use = rec if (rec is not None and not stale) else {}
    traffic = use.get("hbm_bytes_per_launch")
    insts = use.get("SQ_INSTS_VALU")
    floor_ms = compulsory_bytes(m, n, T, K) / (HBM_PEAK_GBS * 1e9) * 1e3
    alg_gbs = alg_bytes / (sl_ms * 1e-3) / 1e9
    hbm = {
        "alg_bytes_per_launch": alg_bytes, "alg_achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "alg_frac": alg_gbs / HBM_PEAK_GBS,
        "alg_frac_note": "the contract's figure (algorithmic bytes / time / 8 TB/s); saturated for a kernel that serves its taps "
                         "from LDS - north_star's 40 % target is met by construction, read hbm_frac and kernel_over_floor",
        "hbm_frac": (traffic / (sl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "floor_ms": floor_ms, "kernel_over_floor": sl_ms / floor_ms,
    }
    rf = {"kernel": kernel, "kernel_ms": sl_ms, "traffic": traffic, "traffic_stale": bool(stale), "hbm": hbm,
          "counter_source": rec.get("source") if rec else None}
    if choice == 12:
        achieved = (insts / (sl_ms * 1e-3) / 1e9) if insts else None
        rf.update({"bound": "valu issue", "achieved": achieved, "peak": VALU_PEAK_GINST, "unit": "G wave64 VALU instructions/s",
                   "frac": (achieved / VALU_PEAK_GINST) if achieved else None,
                   "valu_insts_per_launch": insts,
                   "frac_at_measured_clock": (insts * 4.0 / (N_SIMD * use["GRBM_GUI_ACTIVE"] / 8.0))
                   if insts and use.get("GRBM_GUI_ACTIVE") else None,
                   "frac_note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x cycles): `frac` takes the cycles of kernel_ms at the "
                                "2.4 GHz maximum, frac_at_measured_clock the GRBM_GUI_ACTIVE / 8 of the counter pass itself"})
    else:
        # the gather kernels move their taps through the vector-memory path: the contract's HBM reading applies
        rf.update({"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS})
    return rf


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


def steps_loop_workload(precip_d, vel_d, n_members, first_member, n_total, T, K, levels=6, download=False):
    """BASELINE config 4 on one rank as nowcasts.steps runs it: the member update of
    pysteps/nowcasts/steps.py:1057-1219 (white noise from the member's MT19937 stream -> noise filter ->
    cascade decomposition -> AR(2) step per level -> recomposition -> incremental mask -> CDF matching ->
    mask update) followed by the member-batched advection with BPS-perturbed motion
    (pysteps/nowcasts/utils.py:441-462), for `n_members` of `n_total` members (global indices
    first_member ...: their slice of the ensemble's seed chain, steps.py:885-933), everything resident in
    HBM.  The initial state is synthetic (cascade of the input field, Gaussian band-pass weights, AR(2)
    parameters from lag correlations 0.9^(k+1)); the work per member and lead time does not depend on it.
    Returns (step, info): step() = one nowcast of T lead times = T + 1 updates and T advections
    (download=True: plus one device-to-host copy of the advected members per lead time)."""
    from pysteps_amd import _lib, _pinned
    from pysteps_amd.cascade import decomposition_fft, recompose_fft
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector, steps_noise_generators, steps_perturbators
    from pysteps_amd.noise.fftgenerators import generate_noise_2d_fft_filter
    from pysteps_amd.nowcasts.steps_resident import ResidentSteps
    from pysteps_amd.nowcasts.utils import compute_dilated_mask

    lib = _lib.lib()
    m, n = precip_d.shape
    plane, p = m * n, 2
    field = DeviceArray((m, n), np.float64)
    _lib.check(lib.psh_convert_dev(precip_d.ptr, field.ptr, plane, 1), "psh_convert_dev")
    host = field.to_host()
    # the latest "observation": radar products carry ONE no-rain value; the advected input frame has it
    # smeared over a few float32 neighbours of -15 by the interpolation
    target = DeviceArray.from_host(np.where(host < -14.5, -15.0, host))
    # band-pass weights of the reference's form (levels, m, n/2+1), Gaussian in log2 of the wavenumber
    ky, kx = np.fft.fftfreq(m)[:, None], np.fft.rfftfreq(n)[None, :]
    r = np.log2(np.hypot(ky, kx) + 1e-9)
    centres = np.log2(0.5 * 2.0 ** -np.arange(levels)[::-1])
    w = np.stack([np.exp(-0.5 * ((r - c) / 0.6) ** 2) for c in centres])
    w[0][r <= centres[0]] = 1.0  # the largest scales (the mean included) belong to the first level,
    w[-1][r >= centres[-1]] = 1.0  # the smallest to the last: every wavenumber has weight
    bp = {"weights_2d": w / w.sum(axis=0, keepdims=True), "weights_1d": np.zeros((levels, 1)), "shape": (m, n)}
    dec = decomposition_fft(field, bp, normalize=True, compute_stats=True)
    cascades = DeviceArray((n_members, levels, p, m, n), np.float64)
    for j in range(n_members):
        for k in range(levels):
            for slot in range(p):
                dst = cascades.ptr + (((j * levels + k) * p + slot) * plane) * 8
                _lib.check(lib.psh_memcpy_d2d(dst, dec["cascade_levels"].ptr + k * plane * 8, plane * 8), "d2d")
    # AR(2) by Yule-Walker from lag-1 / lag-2 correlations (timeseries/autoregression.py estimate_ar_params_yw)
    phi = np.empty((levels, 3))
    for k in range(levels):
        r1 = 0.9 ** (k + 1)
        r2 = r1 ** 2 * 0.98
        a1, a2 = r1 * (1.0 - r2) / (1.0 - r1 * r1), (r2 - r1 * r1) / (1.0 - r1 * r1)
        phi[k] = (a1, a2, np.sqrt(max(1.0 - a1 * r1 - a2 * r2, 1e-6)))
    spec = np.abs(np.fft.rfft2(host - host.mean()))
    noise_filter = {"field": spec / spec.std(), "input_shape": (m, n), "use_full_fft": False}
    thr = -10.0
    wet = DeviceArray((m, n), np.uint8)
    _lib.check(lib.psh_ge_mask_dev(field.ptr, plane, thr, wet.ptr), "psh_ge_mask_dev")
    from scipy.ndimage import generate_binary_structure

    struct = generate_binary_structure(2, 1)
    grey0 = compute_dilated_mask(wet, struct, 10)
    grey = DeviceArray((n_members, m, n), np.float64)
    for j in range(n_members):
        _lib.check(lib.psh_memcpy_d2d(grey.view(j).ptr, grey0.ptr, plane * 8), "d2d")
    decomp = [{"means": dec["means"], "stds": dec["stds"], "normalized": True, "domain": "spatial"} for _ in range(n_members)]
    gens = steps_noise_generators(n_total, 42)[first_member:first_member + n_members]
    state = {"randgen_prec": gens, "precip_cascades": cascades, "precip_decomp": decomp, "mask_prec": grey}
    params = {"noise_method": "nonparametric", "domain": "spatial", "generate_noise": generate_noise_2d_fft_filter,
              "pert_gen": noise_filter, "decomp_method": decomposition_fft, "recomp_method": recompose_fft, "filter": bp,
              "phi": phi, "noise_std_coeffs": np.ones(levels), "n_cascade_levels": levels, "n_ens_members": n_members,
              "mask_method": "incremental", "probmatching_method": "cdf", "precip": target, "precip_thr": thr,
              "domain_mask": None, "struct": struct, "mask_rim": 10}
    loop = ResidentSteps(state, params, (m, n), 1 << 40)
    timestep_min = 5.0
    perts = steps_perturbators(n_total, 42, 1.0, timestep_min)[first_member:first_member + n_members]
    adv = EnsembleAdvector(vel_d, n_members, perts, n_iter=K, outval=-15.0)
    block = _pinned.empty((n_members, T, m, n), np.float64) if download else None
    f32 = DeviceArray((n_members, m, n), np.float32)
    copies = []  # (event before, event after) of every lead time's device-to-host copies

    def step(events=None):
        adv.reset()
        loop.update()  # the update of t = 0 (nowcasts/utils.py:386-395: func is called before the first lead time)
        out = None
        for t in range(T):
            new = loop.update()
            _lib.check(lib.psh_convert_dev(new.ptr, f32.ptr, new.size, 0), "psh_convert_dev")
            if events is not None:
                events[t][0].record()
            out = adv.step(f32, 1.0, timestep_min * (t + 1))
            if events is not None:
                events[t][1].record()
            if block is not None:
                from pysteps_amd.device import Event  # noqa: PLC0415

                c0 = Event().record()
                wide = DeviceArray(out.shape, np.float64)
                _lib.check(lib.psh_convert_dev(out.ptr, wide.ptr, out.size, 1), "psh_convert_dev")
                for j in range(n_members):
                    _lib.check(lib.psh_memcpy_d2h_async(block[j, t].ctypes.data, wide.ptr + j * plane * 8, plane * 8), "d2h")
                copies.append((c0, Event().record()))
        return out

    info = {"members": n_members, "cascade_levels": levels, "ar_order": p, "mask_method": "incremental",
            "probmatching_method": "cdf", "updates_per_nowcast": T + 1, "advections_per_nowcast": T}
    step.copies = copies
    return step, info


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
           "irfft2_hbm_frac": traffic / (inv * 1e-3) / 1e9 / HBM_PEAK_GBS, "fft_alg_bytes": traffic,
           "fft_alg_bytes_note": "two-pass transform: real plane once, half spectrum three times (written by the first "
                                 "pass, read and written by the second)",
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
        w[0][r <= centres[0]] = 1.0
        w[-1][r >= centres[-1]] = 1.0
        bp = {"weights_2d": w / w.sum(axis=0, keepdims=True), "weights_1d": np.zeros((nlev, 1)), "shape": (m, n)}
        decomposition_fft(dx, bp, normalize=True, compute_stats=True)  # uploads and caches the weights
        synchronize()
        e3, e4 = Event(), Event()
        e3.record()
        decomposition_fft(dx, bp, normalize=True, compute_stats=True)
        e4.record()
        synchronize()
        ms = e3.elapsed_ms(e4)
        out["cascade_decompose_%d_levels_ms" % nlev] = ms
        # per pixel: field in 8 + half spectrum out 8; per level: spectrum read 8 + weights read 4 + level written 8,
        # moments read 8, normalisation read + write 16
        cas_bytes = (16.0 + nlev * 44.0) * m * n
        out["cascade_alg_bytes"] = cas_bytes
        out["cascade_hbm_frac"] = cas_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
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
        pm_ms = e5.elapsed_ms(e6) / reps
        out["probmatch_cdf_ms"] = pm_ms
        # round 4: the forecast's side runs without device-scope atomics (two partition passes with LDS histograms,
        # docs/history.md 3.8); what bounds it now is bytes - the wet pixels travel as 16-byte records through two
        # scatters and the ranked values go back to their pixels as scattered 8-byte stores
        out["probmatch_bound"] = {"bound": "hbm", "min_bytes_per_call": 24.0 * m * n,
                                  "hbm_frac_of_min_bytes": 24.0 * m * n / (pm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "min bytes: forecast read, observation read, result written once (8 B each per pixel); "
                                          "this call also ranks the observation - the member loop keeps that in a plan"}
    except Exception as exc:
        out["probmatch_note"] = "%s: %s" % (type(exc).__name__, exc)
    return out


def steps_loop_leg(precip_d, vel_d, members, T, K, dist, with_stock):
    """config.steps_loop / config.steps_e2e of the N = 1 line: the per-GPU share of BASELINE config 4
    (`members` STEPS members, T lead times) as one GPU runs it - resident (value-style, nothing leaves
    HBM) and end to end with one device-to-host copy of the advected members per lead time - and, as
    the CPU figure beside it, the reference's own nowcasts.steps (oracle/_ref) on a stated sample."""
    from pysteps_amd.device import Event, synchronize

    m, n = precip_d.shape
    step, info = steps_loop_workload(precip_d, vel_d, members, 0, members, T, K)
    ev = [(Event(), Event()) for _ in range(T)]
    step()
    synchronize()
    t0 = time.perf_counter()
    step(ev)
    synchronize()
    resident_s = time.perf_counter() - t0
    adv_ms = sum(a.elapsed_ms(b) for a, b in ev) / T
    out = dict(info)
    out.update({"leadtimes": T, "seconds_per_nowcast": resident_s, "ms_per_leadtime_all_members": resident_s / T * 1e3,
                "ms_per_member_update": (resident_s * 1e3 - adv_ms * T) / (members * (T + 1)),
                "advection_ms_per_leadtime": adv_ms, "value": members * m * n * T / resident_s / 1e6,
                "unit": "Mpx*leadsteps/s", "note": "state resident in HBM, results stay on the device"})
    # roofline of the member update (HBM): algorithmic bytes = every array of the update read or written once
    # per stage that has to see it whole - L (p + 1) half spectra of the AR history, white field, noise
    # spectrum, recomposed spectrum, field, masked field (each written + read), matched field written, mask
    # read + written: 8 (L (p + 1) + 13) B per pixel (docs/history.md 3.12); `traffic` is the PMC measurement
    levels, order = int(info["cascade_levels"]), int(info["ar_order"])
    alg = 8.0 * (levels * (order + 1) + 13) * m * n
    upd_s = out["ms_per_member_update"] * 1e-3
    traffic, stale = None, False
    try:
        with open(os.path.join(ROOT, "profiles", "member_update_traffic.json")) as fh:
            rec = json.load(fh)
        if rec.get("workload", "").startswith("%dx%d" % (m, n)):
            # (counters of another revision of the update's kernels are withheld: tools/member_traffic.py stores the hashes)
            stale = not sources_match(rec.get("source_hashes"), MU_SOURCES)
            traffic = None if stale else rec["hbm_bytes_per_member_update"]
    except Exception:
        stale = False
    out["roofline"] = {"kernel": "member update (all kernels of one update)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                       "alg_bytes_per_member_update": alg, "achieved": alg / upd_s / 1e9, "frac": alg / upd_s / 1e9 / HBM_PEAK_GBS,
                       "traffic": traffic, "traffic_stale": bool(stale),
                       "hbm_frac": (traffic / upd_s / 1e9 / HBM_PEAK_GBS) if traffic else None}
    del step
    e2e = None
    try:
        step, _ = steps_loop_workload(precip_d, vel_d, members, 0, members, T, K, download=True)
        step()
        synchronize()
        e0, e1 = Event(), Event()
        t0 = time.perf_counter()
        e0.record()
        step()
        e1.record()
        synchronize()
        wall = time.perf_counter() - t0
        moved = members * T * m * n * 8.0
        copy_s = sum(a.elapsed_ms(b) for a, b in step.copies[-T:]) * 1e-3
        e2e = {"device_seconds": wall, "bytes_to_host": moved, "transfer_seconds": copy_s,
               "transfer_share": copy_s / wall if wall > 0 else None, "transfer_gb_per_s": moved / copy_s / 1e9 if copy_s else None,
               "note": "same loop with one device-to-host copy of the advected members (widened to float64 on the device, "
                       "pinned result block) per lead time; transfer_seconds by HIP events around the copies"}
    except Exception as exc:  # this leg never takes the headline down
        e2e = {"note": "%s: %s" % (type(exc).__name__, exc)}
    if with_stock:
        e2e["stock"] = steps_stock_sample(m, n)
        if e2e.get("device_seconds") and e2e["stock"].get("loop_seconds_per_member_update"):
            full = e2e["stock"]["loop_seconds_per_member_update"] * members * (T + 1)
            e2e["stock"]["loop_seconds_scaled_to_this_nowcast"] = full
            e2e["speedup_of_the_loop"] = full / e2e["device_seconds"]
    e2e["real_caller"] = ("the REAL pysteps.nowcasts.steps driving this loop (register(patch_main_loop=True)) is timed by "
                          "tools/steps_quick.py: profiles/r03/*_steps_quick.jsonl")
    return out, e2e


def steps_stock_sample(m, n):
    """The reference's own nowcasts.steps (oracle/_ref, stock operators) on 1 member x 1 lead time of the
    same grid: initialisation and loop seconds on this box's host (1 core: NumPy / SciPy are
    single-threaded here).  CPU baseline only - never part of `value`."""
    try:
        import contextlib
        import io

        from oracle import build_ref

        if not build_ref.available():
            return {"note": "oracle/_ref not built"}
        build_ref.activate()
        from pysteps import nowcasts
        from tools import synth

        frames = synth.steps_frames(m, n, 3)
        V = synth.true_velocity(m, n).astype(np.float64)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            _, init_s, loop_s = nowcasts.get_method("steps")(
                frames, V, 1, n_ens_members=1, n_cascade_levels=6, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, seed=42,
                vel_pert_method="bps", mask_method="incremental", probmatching_method="cdf", num_workers=1, measure_time=True)
        return {"kind": "reference", "cores": 1, "sample": "pysteps.nowcasts.steps (oracle/_ref, unmodified), %dx%d, 1 member x 1 lead "
                "time (2 member updates + 1 advection), stock operators" % (m, n), "total_seconds": time.perf_counter() - t0,
                "init_seconds": init_s, "loop_seconds": loop_s, "loop_seconds_per_member_update": loop_s / 2.0}
    except Exception as exc:
        return {"note": "%s: %s" % (type(exc).__name__, exc)}


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

    if args.workload == "config5":
        return main_config5(args, dist)
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
        "roofline": None,
    }
    line["roofline"] = semilag_roofline(m, n, T, K, sl_ms, alg_bytes)
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
    if not args.no_steps_loop:
        try:
            loop_T = min(T, 6)
            sl, e2e = steps_loop_leg(precip_d, vel_d, args.members_per_gpu, loop_T, K, dist,
                                     with_stock=not (args.no_cpu_baseline or args.no_steps_stock))
            line["config"]["steps_loop"] = sl
            line["config"]["steps_e2e"] = e2e
        except Exception as exc:  # never takes the headline down
            line["config"]["steps_loop"] = {"note": "%s: %s" % (type(exc).__name__, exc)}
    # the three figures every line carries under the same keys (see main_members): this line's `value` is
    # lk_semilag_value (BASELINE config 3); the other two are what one GPU of an N > 1 run does
    line["lk_semilag_value"] = value
    line["advection_value"] = line["config"].get("config4_one_gpu", {}).get("value")
    line["member_loop_value"] = line["config"].get("steps_loop", {}).get("value")
    line["config"]["workload"] += "; `value` = lk_semilag_value"
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(frames_d, vel_d, K, args.cpu_sample_steps, T, have_lk)
    print(json.dumps(line))
    dist.close()


def members_pmc_traffic(m, n, members):
    """HBM bytes per launch of semilag_members from the committed counter passes (profiles/r03, taken
    with tools/gpu_members_round.sh at 4096^2 x 6 members; FETCH_SIZE doubled per the gfx950 note)."""
    if (m, n, members) != (4096, 4096, 6):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r03", "a_members_pmc_traffic.json")) as fh:
            t = json.load(fh)
        return (2.0 * t["FETCH_SIZE_KB_mean_per_launch"] + t["WRITE_SIZE_KB_mean_per_launch"]) * 1024.0
    except Exception:
        return None


def main_members(args, dist, dense_lk):
    """N > 1 (one rank per GPU): BASELINE config 4, `members_per_gpu` STEPS members per GPU.

    Rank 0 synthesises the frames, estimates the motion field and packs [precip | u | v] into one
    buffer; it then times ITS per-GPU workload alone (the other ranks wait at a barrier): the in-run
    single-GPU base of the weak-scaling figure.  ONE RCCL broadcast over xGMI (192 MiB at 4096^2)
    hands the buffer to every rank before the timed region; each rank runs the member loop of
    nowcasts.steps for its own members (partition of 6N members; random streams and perturbators
    recomputed from the ensemble's seed chain; member update + advection per lead time, everything
    resident) with no data-path collective (weak scaling).  --advection-only times the advection share
    alone (the surrogate of rounds 1-2).  An RCCL failure is fatal: the run exits non-zero."""
    from pysteps_amd import _lib, parallel
    from pysteps_amd.device import DeviceArray, Event, synchronize

    m = n = args.size
    T, K = args.leadtimes, args.n_iter
    per = args.members_per_gpu
    n_total = per * dist.world
    mine = parallel.partition(n_total, dist.world, dist.rank)
    pack = DeviceArray((3, m, n), np.float32)
    precip_d = pack.view(0)
    vel_d = DeviceArray((2, m, n), np.float32, ptr=pack.view(1).ptr, owner=pack)

    def build():
        if args.advection_only:
            return members_workload(precip_d, vel_d, len(mine), mine.start, n_total, T, K), None
        return steps_loop_workload(precip_d, vel_d, len(mine), mine.start, n_total, T, K)

    step = info = None
    base = None
    if dist.rank == 0:
        frames_d, v_true = make_inputs(m, n, args.frames)
        v = dense_lk(frames_d) if dense_lk is not None else v_true
        lib = _lib.lib()
        _lib.check(lib.psh_memcpy_d2d(pack.view(0).ptr, frames_d.view(args.frames - 1).ptr, m * n * 4), "d2d")
        _lib.check(lib.psh_memcpy_d2d(pack.view(1).ptr, v.ptr, 2 * m * n * 4), "d2d")
        synchronize()
        del frames_d, v_true, v
        # the same per-GPU workload on this rank alone, before any collective
        step, info = build()
        reps = max(1, min(args.steps, 3))
        for _ in range(max(1, min(args.warmup, 2))):
            step()
        synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        synchronize()
        base_ms = (time.perf_counter() - t0) / reps * 1e3
        base = {"ms_per_step": base_ms, "value": len(mine) * m * n * T / (base_ms * 1e-3) / 1e6, "steps": reps,
                "note": "rank 0 alone on its %d members before the collective phase, other ranks idle at a barrier" % len(mine)}
    dist.barrier()
    t0 = time.perf_counter()
    ok = 1.0
    err = None
    comm_init_s = bcast_ms = None
    try:
        with stdout_to_stderr():
            comm = parallel.Communicator(dist.rank, dist.world, dist.broadcast_bytes)
            synchronize()
            comm_init_s = time.perf_counter() - t0  # unique id over the control plane + ncclCommInitRank
            # the data path's ONE collective, by HIP events on the library stream: a first, untimed broadcast brings the
            # rings / channels up (RCCL connects lazily), the second one is the transfer itself
            comm.broadcast(pack, root=0)
            synchronize()  # (the first broadcast also lines the ranks up: no control-plane barrier inside this try block)
            b0, b1 = Event(), Event()
            b0.record()
            comm.broadcast(pack, root=0)
            b1.record()
            synchronize()
            bcast_ms = b0.elapsed_ms(b1)
    except Exception as exc:
        ok, err = 0.0, exc
    if dist.max(1.0 - ok) > 0.0:
        print("rank %d: RCCL broadcast failed on %s: %s" % (
            dist.rank, "this rank" if ok == 0.0 else "another rank", err), file=sys.stderr)
        dist.close()
        sys.exit(3)
    bcast_s = time.perf_counter() - t0
    if step is None:
        step, info = build()

    ev = [[(Event(), Event()) for _ in range(T)] for _ in range(args.steps)]
    if args.advection_only:
        flat = [(Event(), Event()) for _ in range(args.steps)]
        elapsed = time_steps(step, dist, args.steps, args.warmup, flat)
        kernel_ms = sum(a.elapsed_ms(b) for a, b in flat) / args.steps / T
    else:
        for _ in range(args.warmup):
            step()
        synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(ev[i])
        synchronize()
        dist.barrier()
        elapsed = dist.max(time.perf_counter() - t0)
        kernel_ms = sum(a.elapsed_ms(b) for row in ev for a, b in row) / (args.steps * T)  # one batched launch
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * m * n * T / (ms_per_step * 1e-3) / 1e6
    # the same three figures under the same keys in EVERY line (N = 1 included), so that lines of different N
    # can be laid side by side: member_loop_value (update + advection), advection_value (advection alone),
    # lk_semilag_value (BASELINE config 3, N = 1 only).  `value` of this line is member_loop_value
    # (advection_value with --advection-only); its N = 1 twin is `--gpus 1 --force-members-path`
    advection_value = value if args.advection_only else None
    if not args.advection_only:
        try:
            adv_step = members_workload(precip_d, vel_d, len(mine), mine.start, n_total, T, K)
            adv_el = time_steps(adv_step, dist, 2, 1)
            advection_value = n_total * m * n * T / (adv_el / 2) / 1e6
            del adv_step
        except Exception:
            advection_value = None
    # stateful single-step call: D read + write 16, three velocity passes 8 each
    # (increment rebuild, midpoint, end point) for n_iter = 1, field 4, store 4 -> 48 B / px / member
    b_alg = (24 + 16 * K + 8) if K > 0 else 32
    alg_bytes = float(b_alg) * m * n * len(mine)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    if dist.rank == 0:
        what = "member advection only" if args.advection_only else "nowcasts.steps member loop: update + advection per lead time"
        traffic = members_pmc_traffic(m, n, len(mine))
        line = {
            "metric": "Mpixels*leadsteps/s (STEPS ensemble, %s) at %dx%d" % (
                "advection only" if args.advection_only else "member update + semilag advection", m, n),
            "value": value,
            "unit": "Mpx*leadsteps/s",
            "n_gpus": dist.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 advection, f64 member update (the reference's dtypes)",
            "data": "synthetic",
            "member_loop_value": None if args.advection_only else value,
            "advection_value": advection_value,
            "lk_semilag_value": None,
            "single_gpu_base": base,
            "weak_scaling_efficiency": value / (dist.world * base["value"]),
            "config": {
                "workload": "%dx%d, %d-member STEPS ensemble (BASELINE config 4), %d members per GPU, %d lead times, %s, "
                            "BPS velocity perturbations, n_iter=%d; `value` = %s" % (
                                m, n, n_total, per, T, what, K, "advection_value" if args.advection_only else "member_loop_value"),
                "sharding": "members partitioned over ranks (random streams and perturbators from the ensemble's seed chain), "
                            "motion field from %s on rank 0, [precip|u|v] in ONE RCCL broadcast before the timed region, no "
                            "data-path collective" % ("dense LK" if dense_lk is not None else "the synthetic truth"),
                "rccl_ranks": dist.world,
                "comm_init_s": comm_init_s,
                "broadcast_ms": bcast_ms,
                "broadcast_gbs": (pack.nbytes / (bcast_ms * 1e-3) / 1e9) if bcast_ms else None,
                "broadcast_note": "comm_init_s = communicator bring-up (unique id + ncclCommInitRank, host clock, rank 0); "
                                  "broadcast_ms = the [precip | u | v] broadcast alone between two HIP events on the library "
                                  "stream of rank 0 (second of two: RCCL connects its channels on the first); "
                                  "collective_phase_s = both + the warm-up broadcast + barriers",
                "collective_phase_s": bcast_s,
                "broadcast_bytes": pack.nbytes,
                "member_loop": info,
                "compare_with": "single_gpu_base (same run, same per-GPU workload); the --gpus 1 line is BASELINE config 3 "
                                "(LK + semilag of one field) and carries this workload as config.steps_loop / config4_one_gpu",
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
                "traffic": traffic,
                "hbm_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            },
        }
        print(json.dumps(line))
    dist.close()


def main_config5(args, dist):
    """--workload config5 (BASELINE configs[4]): 8192^2, 2 frames, 36 lead times tiled over the ranks.
    Every rank holds the frames (ONE RCCL broadcast before the timed region), runs the whole dense Lucas-Kanade
    estimate itself (deterministic: every rank gets the same field, no collective in the step; --config5-lk banded:
    the image passes on row bands with global ranges by allreduce MIN / MAX / SUM and corner candidates / tracked
    vectors by allgather, pysteps_amd/motion/banded.py) and integrates the rows of its band
    (parallel.tiled_extrapolate, no halo exchange: the scheme has no inter-pixel dependency).  Strong
    scaling: the total work is fixed; rank 0 times the whole step alone first (single_gpu_base)."""
    from pysteps_amd import _lib, extrapolation, motion, parallel
    from pysteps_amd.device import DeviceArray, synchronize

    m = n = args.size if args.size != 4096 else 8192
    T = args.leadtimes if args.leadtimes != 24 else 36
    K = args.n_iter
    frames_d = DeviceArray((args.frames, m, n), np.float32)
    base = None
    if dist.rank == 0:
        made, _ = make_inputs(m, n, args.frames)
        _lib.check(_lib.lib().psh_memcpy_d2d(frames_d.ptr, made.ptr, made.nbytes), "d2d")
        synchronize()
        del made
        dense_lk = motion.get_method("LK")
        extrapolate = extrapolation.get_method("semilagrangian")

        def whole():
            return extrapolate(frames_d.view(args.frames - 1), dense_lk(frames_d), T, outval=-15.0, n_iter=K)

        whole()
        synchronize()
        reps = max(1, min(args.steps, 3))
        t0 = time.perf_counter()
        for _ in range(reps):
            whole()
        synchronize()
        base_ms = (time.perf_counter() - t0) / reps * 1e3
        base = {"ms_per_step": base_ms, "value": m * n * T / (base_ms * 1e-3) / 1e6, "steps": reps,
                "note": "rank 0 alone: single-device dense LK + semilag of the whole grid, before the collective phase"}
    dist.barrier()
    ok, err = 1.0, None
    comm = None
    try:
        with stdout_to_stderr():
            comm = parallel.Communicator(dist.rank, dist.world, dist.broadcast_bytes)
            comm.broadcast(frames_d, root=0)
            synchronize()
    except Exception as exc:
        ok, err = 0.0, exc
    if dist.max(1.0 - ok) > 0.0:
        print("rank %d: RCCL broadcast failed on %s: %s" % (dist.rank, "this rank" if ok == 0.0 else "another rank", err),
              file=sys.stderr)
        dist.close()
        sys.exit(3)
    precip_d = frames_d.view(args.frames - 1)

    from pysteps_amd.device import Event

    marks = []

    plan = parallel.config5_plan(m, dist.world, dist.rank, args.config5_lk)
    dense_lk_whole = motion.get_method("LK")

    def step():
        # (parallel.config5_step, opened up for the event pair around the band's extrapolation)
        v = parallel.banded_dense_lucaskanade(frames_d, comm) if plan["lk"] == "banded" else dense_lk_whole(frames_d)
        e0 = Event().record()
        out = parallel.tiled_extrapolate(precip_d, v, T, dist.rank, dist.world, outval=-15.0, n_iter=K)
        marks.append((e0, Event().record()))
        return out

    try:
        with stdout_to_stderr():
            elapsed = time_steps(step, dist, args.steps, args.warmup)
    except Exception as exc:
        print("rank %d: config-5 step failed: %s" % (dist.rank, exc), file=sys.stderr)
        dist.close()
        sys.exit(3)
    ms_per_step = elapsed / args.steps * 1e3
    value = m * n * T / (ms_per_step * 1e-3) / 1e6
    if dist.rank == 0:
        rows = parallel.partition(m, dist.world, 0)
        sl_ms = sum(a.elapsed_ms(b) for a, b in marks[-args.steps:]) / args.steps
        alg_bytes = float((16 * K + 8) if K > 0 else 16) * len(rows) * n * T
        line = {
            "metric": "Mpixels*leadsteps/s (LK+semilag) at %dx%d fp32, row bands over %d GPUs" % (m, n, dist.world),
            "value": value, "unit": "Mpx*leadsteps/s", "n_gpus": dist.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "single_gpu_base": base, "strong_scaling_speedup": value / base["value"],
            "strong_scaling_efficiency": value / base["value"] / dist.world,
            "config": {
                "workload": "%dx%d fp32, %d input frames, %s dense LK + tiled semilag %d leadtimes n_iter=%d (BASELINE "
                            "config 5)" % (m, n, args.frames, plan["lk"], T, K),
                "sharding": "row bands of the extrapolation: %d rows per rank, no halo exchange; frames in ONE RCCL broadcast "
                            "before the timed region; motion estimate %s" % (len(rows), (
                                "replicated on every rank (deterministic: bit-identical fields, NO collective in the step)"
                                if plan["lk"] == "replicated" else
                                "in row bands: per estimate allreduce MIN/MAX/SUM of <= 6 floats x 3, allgather of corner "
                                "keys and of tracked vectors")),
                "config5_plan": plan,
                "rccl_ranks": dist.world,
            },
            "roofline": {"kernel": "semilag_window (row band of rank 0)", "bound": "hbm", "achieved": alg_bytes / (sl_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (sl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": alg_bytes, "kernel_ms": sl_ms, "traffic": None,
                         "note": "kernel_ms brackets the band's extrapolation call (velocity / field packing passes included)"},
            "config_lk_ms_per_step": ms_per_step - sl_ms,
        }
        print(json.dumps(line))
    dist.close()


if __name__ == "__main__":
    main()
