#!/bin/bash
# Stand-alone durations of the LK kernels (no side-stream overlap): rocprofv3 kernel statistics of
# tools/lk_quick.py, whose stages run one after the other with a synchronisation in between.
# Usage: bash tools/gpu_lk_standalone.sh <tag> [ENV=VALUE ...]
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quick -- python tools/lk_quick.py 4096 > $OUT/quick.log 2>&1
f=$(find $OUT/quick -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/standalone.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    print("%-70s calls %4s avg %9.1f us  min %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
find $OUT/quick -name "*kernel_trace.csv" -delete; find $OUT/quick -name "*agent_info.csv" -delete
