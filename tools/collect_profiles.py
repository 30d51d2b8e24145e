"""Copy the judged summaries of a gpurun_out/<tag> session into profiles/<round>/ and refresh
profiles/pmc_traffic.json (HBM bytes per launch of the extrapolation kernel; FETCH_SIZE is doubled: gfx950
reports exactly half of a coalesced read stream, see tools/calib_copy.py / DESIGN.md section 6)."""
import csv, glob, hashlib, json, os, shutil, sys


def blob_hash(path):
    """git's blob hash of a working-tree file: bench.py compares these with the tree it runs from and withholds the
    counters (`traffic_stale` / `roofline_lk.stale`) when a kernel source changed after they were taken"""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


SL_SOURCES = ("pysteps_amd/csrc/semilag.hip", "pysteps_amd/csrc/semilag_device.h", "pysteps_amd/csrc/common.h")
LK_SOURCES = ("pysteps_amd/csrc/lk.hip", "pysteps_amd/csrc/lk_sparse.hip", "pysteps_amd/csrc/sparse_qc.hip",
              "pysteps_amd/csrc/idw.hip", "pysteps_amd/csrc/dense_lk.hip", "pysteps_amd/csrc/common.h")

tag, rnd, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join("gpurun_out", tag)
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
for name in ("bench.json", "pytest_gpu.txt", "gaps.txt", "lk_timeline.txt", "fft_quick.json", "other_shapes.jsonl",
             "bench_members_world1.json", "bench_members_advection_world1.json", "bench_config5_world1.json",
             "bench_config5_banded_world1.json", "pytest_window_forced.txt", "debug_build.txt",
             "steps_quick.jsonl", "rng_quick.json", "ensemble_quick.txt", "cv2_probe.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (prefix, name)))
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "%s_rocprofv3_kernel_stats.csv" % prefix))

# per-kernel averages of the traced bench run (5 timed + 2 warm-up steps) for bench.py's roofline_lk
STEPS_TRACED = 7
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    # keyed by the kernel's base name AND, for templates, by every instantiation (round 3 mixed the one-step
    # input synthesis launch of semilag_fused<..., 0, ...> into the averages of the 24-lead-time kernel);
    # the base-name entry of a template is its instantiation with the largest total time
    table, inst = {}, {}
    for r in csv.DictReader(open(f)):
        full = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("psh::", "").strip()
        if full.startswith("void "):
            full = full[5:]
        rec = {"calls": int(r["Calls"]), "total_ns": float(r["TotalDurationNs"])}
        rec["avg_ns"] = rec["total_ns"] / max(rec["calls"], 1)
        rec["ns_per_step"] = rec["total_ns"] / STEPS_TRACED
        inst[full] = rec
    for full, rec in inst.items():
        base = full.split("<")[0]
        if base not in table or rec["total_ns"] > table[base]["total_ns"]:
            table[base] = dict(rec, instantiation=full)
    for full, rec in inst.items():
        if "<" in full:
            table[full] = rec
    # per-kernel counters of the PMC passes of the same command (tools/pmc_passes.sh summary), if taken
    counters, counter_source = {}, None
    summary = os.path.join(src, "pmc", "summary.csv")
    if os.path.exists(summary):
        for line in list(open(summary))[1:]:
            # kernel names carry commas (template arguments): the two last fields are counter and value
            kernel, counter, value = line.rstrip("\n").rsplit(",", 2)
            kernel, counter = kernel.strip(), counter.replace("_sum", "")
            counters.setdefault(kernel, {})[counter] = float(value)  # per instantiation (names cut at 40 characters)
        for k in list(counters):
            base = k.split("<")[0]
            chosen = table.get(base, {}).get("instantiation", base)[:40].strip()
            if "<" in k and k == chosen:
                counters[base] = dict(counters[k])  # the base name carries its dominant instantiation's counters
        shutil.copy(summary, os.path.join(dst, "%s_pmc_kernels.csv" % prefix))
        counter_source = "profiles/%s/%s_pmc_kernels.csv" % (rnd, prefix)
    json.dump({"source": "profiles/%s/%s_rocprofv3_kernel_stats.csv" % (rnd, prefix), "steps_traced": STEPS_TRACED,
               "workload": "4096x4096", "counters": counters, "counter_source": counter_source,
               "source_hashes": {src: blob_hash(src) for src in LK_SOURCES},
               "note": "ns_per_step also spreads the input synthesis launches of semilag_fused over the steps; "
                       "LK kernels only run inside steps", "kernels": table},
              open(os.path.join("profiles", "kernel_stats_latest.json"), "w"), indent=1)


def mean_counter(sub, counter, kernel):
    vals = []
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    vals = [v for v in vals if v[1] > 500000]  # full-size launches only (not the input synthesis)
    return (sum(v[0] for v in vals) / len(vals), sum(v[1] for v in vals) / len(vals) / 1e6, len(vals)) if vals else None

SL_KERNEL = "semilag_window"  # the extrapolator of the bench step (rounds 1-4: semilag_fused)
fetch = mean_counter("pmc_fetch", "FETCH_SIZE", SL_KERNEL)
write = mean_counter("pmc_write", "WRITE_SIZE", SL_KERNEL)
if fetch and write:
    bench = json.load(open(os.path.join(src, "bench.json")))
    key = "semilag_4096x4096_T24_K1"
    rec = {
        "kernel": SL_KERNEL, "launches_averaged": fetch[2],
        "FETCH_SIZE_KiB": fetch[0], "WRITE_SIZE_KiB": write[0],
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": (2.0 * fetch[0] + write[0]) * 1024.0,
        "kernel_ms_under_pmc": fetch[1],
        "alg_bytes_per_launch": (bench["roofline"].get("hbm") or bench["roofline"])["alg_bytes_per_launch"],
        "source": "gpurun_out/%s pmc_fetch + pmc_write (rocprofv3 --pmc, separate passes)" % tag,
        "source_hashes": {src: blob_hash(src) for src in SL_SOURCES},
    }
    # VALU instructions and chip cycles of the same launch (the roofline that binds the window kernel: VALU issue)
    stats = json.load(open(os.path.join("profiles", "kernel_stats_latest.json"))) if os.path.exists(
        os.path.join("profiles", "kernel_stats_latest.json")) else {}
    c = stats.get("counters", {}).get(SL_KERNEL, {})
    for name in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_LDS"):
        if name in c:
            rec[name] = c[name]
    if "SQ_INSTS_VALU" in c:
        rec["source"] += "; SQ_* / GRBM_* from %s" % stats.get("counter_source")
    path = os.path.join("profiles", "pmc_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[key] = rec
    json.dump(table, open(path, "w"), indent=1)
    json.dump(rec, open(os.path.join(dst, "%s_pmc_semilag.json" % prefix), "w"), indent=1)
    print(rec)
