#!/bin/bash
# HBM traffic of the probability-matching kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (kernel trace only) over tools/probmatch_quick.py 4096, masked case.  Output: gpurun_out/pm_pmc/summary.csv
export TMPDIR=/tmp PM_CASE=masked
OUT=$PWD/gpurun_out/pm_pmc
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python $OLDPWD/tools/probmatch_quick.py 4096 > $OUT/$c.log 2>&1)
done
python - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv")) + sorted(glob.glob(out + "/*/*counter_collection.csv")):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pm_" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("psh::", "")
        by[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in by.items():
        agg[k] = sum(v) / len(v)
with open(out + "/summary.csv", "w") as fh:
    fh.write("kernel,counter,mean_per_launch\n")
    for (k, c), v in agg.items():
        fh.write("%s,%s,%.6g\n" % (k, c, v))
print(open(out + "/summary.csv").read())
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
