#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel trace + PMC passes.
# Usage (from the repo root, via gpurun):  bash tools/gpu_round.sh <tag>
set -u
TAG=${1:-dev}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 2 2>$OUT/bench.err | tee $OUT/bench.json
BENCH="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- python tools/calib_copy.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -- python tools/calib_copy.py > $OUT/calib_write.log 2>&1
find $OUT -name "*.csv" | head -40
# keep the merged output small: per-dispatch traces can be large
find $OUT -name "*kernel_trace.csv" -size +4M -delete
du -sh $OUT
