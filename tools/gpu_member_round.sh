#!/bin/bash
# Evidence for the STEPS member update (bench.py's steps_loop leg: 6 members, 6 lead times, 4096^2, the
# default - spectral - update): per-kernel statistics, the kernels of ONE member update in launch order,
# HBM bytes (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) and SQ counters per kernel.
# Usage: bash tools/gpu_member_round.sh <tag>
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-lk --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-stock"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mtrace -- $B > $OUT/mtrace.log 2>&1
grep -o '"steps_loop": {[^}]*}' $OUT/mtrace.log | head -1 > $OUT/member_steps_loop.json
python tools/gap_update.py $OUT/mtrace > $OUT/member_update_kernels.txt 2>&1
f=$(find $OUT/mtrace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/member_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/mpmc_$c -- $B > $OUT/mpmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/mpmc_SQ -- $B > $OUT/mpmc_SQ.log 2>&1
python - "$OUT" <<'PY' | tee $OUT/member_pmc.csv
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/mpmc_*/*/*counter_collection.csv")):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("psh::", "")
        by[(name[:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in by.items():
        agg[k] = (sum(v) / len(v), len(v))
print("kernel,counter,mean_per_launch,launches")
for (k, c), (v, n) in agg.items():
    print("%s,%s,%.6g,%d" % (k, c, v, n))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cat $OUT/member_update_kernels.txt | head -60
