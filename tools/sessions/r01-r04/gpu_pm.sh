#!/bin/bash
# Probability matching on the GPU box: parity tests, then the kernel statistics of the masked 4096^2 case
# (the member loop's case) and the resident / host-path timings of both cases.
# Usage: bash tools/gpu_pm.sh <tag>
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_probmatch_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
PM_CASE=masked timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pm -- python tools/probmatch_quick.py 4096 > $OUT/pm_masked.json 2> $OUT/pm.err
f=$(find $OUT/pm -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/pm_kernels_masked.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = 0.0
for r in rows[:26]:
    print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
timeout 400 python tools/probmatch_quick.py 4096 | tee $OUT/pm_quick.json | cut -c1-700
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
