#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group, --kernel-trace only) on a command.
# Usage: bash tools/pmc_passes.sh <outdir> <command...>
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1
done < ${PMC_GROUPS:-tools/pmc_groups_default.txt}
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/*/*counter_collection.csv")):
    keep = ("semilag", "idw", "lk_", "corner_", "vectors_finish", "outliers", "pack_", "fft_", "moments", "standardise")
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in keep)]
    by = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("psh::", "")
        by[(name[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in by.items():
        v = v[len(v)//2:]  # steady-state launches
        agg[k] = sum(v) / len(v)
with open(out + "/summary.csv", "w") as fh:
    fh.write("kernel,counter,mean_per_launch\n")
    for (k, c), v in agg.items():
        fh.write("%s,%s,%.6g\n" % (k, c, v))
print(open(out + "/summary.csv").read())
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
