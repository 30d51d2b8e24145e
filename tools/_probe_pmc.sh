cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/gather_probe.py 2.4e9 short > gpurun_out/probe_r2_timing.txt 2>&1
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d gpurun_out/probe_pmc -- python tools/gather_probe.py 2.4e9 short > gpurun_out/probe_r2_pmc.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/probe_pmc/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
by = collections.OrderedDict()
for r in rows:
    if "calib_gather" not in r["Kernel_Name"]: continue
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
out = open("gpurun_out/probe_r2_pmc.txt", "w")
for d, c in by.items():
    out.write("%s %s\n" % (d, " ".join("%s=%.4g" % kv for kv in sorted(c.items()))))
PY
cat gpurun_out/probe_r2_timing.txt; cat gpurun_out/probe_r2_pmc.txt
