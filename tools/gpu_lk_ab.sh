#!/bin/bash
# One GPU session for LK kernel work: fingerprints against a base, then the per-kernel statistics of
# the bench step for each value of one variant knob.
# Usage: bash tools/gpu_lk_ab.sh <tag> <ENV_VAR> "<values>"   (values: space-separated, "-" = unset)
set -u
TAG=$1; VAR=${2:-NONE}; VALS=${3:--}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/lk_bitcheck.py $TAG > $OUT/bitcheck.log 2>&1
tail -2 $OUT/bitcheck.log
cp profiles/r04/j_lk_bitcheck_base_idw_fma.json gpurun_out/lk_bitcheck_base.json; if [ -f gpurun_out/lk_bitcheck_base.json ]; then python tools/lk_bitcheck.py --diff base $TAG | tee $OUT/bitcheck_diff.txt | head -40; fi
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop"
for v in $VALS; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- $BENCH > $OUT/trace_$v.log 2>&1
  f=$(find $OUT/trace_$v -name "*kernel_stats.csv" | head -1)
  echo "== $VAR=$v"; grep -o '"lk_ms_per_step": [0-9.]*' $OUT/trace_$v.log | head -1
  python - "$f" <<'PY' | tee $OUT/stats_$v.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:32]:
    print("%-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find $OUT/trace_$v -name "*kernel_trace.csv" -delete; find $OUT/trace_$v -name "*agent_info.csv" -delete
done
