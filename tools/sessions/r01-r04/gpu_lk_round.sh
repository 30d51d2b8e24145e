#!/bin/bash
# One GPU-box session for the LK sparse stage and the spectral pieces: their parity tests, a bench
# line without the CPU legs, the rocprofv3 kernel trace of the same command and the idle-gap analysis
# of one step, the host-side timeline + corner walk statistics of one estimate.
# Usage (from the repo root, via gpurun):  bash tools/gpu_lk_round.sh <tag>
set -u
TAG=${1:-lk}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lk_sparse_gpu.py tests/test_fft_gpu.py tests/test_cascade_gpu.py tests/test_lk_gpu.py \
  tests/test_lk_banded_gpu.py tests/test_idw_gpu.py tests/test_callers_gpu.py -q 2>&1 | tail -40 | tee $OUT/pytest_lk.txt
timeout 300 python tools/fft_quick.py 2048 4096 2>&1 | tail -3 | tee $OUT/fft_quick.json
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral"
timeout 300 $BENCH 2>$OUT/bench.err | tee $OUT/bench_quick.json
PYSTEPS_HIP_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral 2>&1 >/dev/null | grep dense_lk | tail -8 | tee $OUT/lk_timeline.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
python tools/gap_analysis.py $OUT/trace > $OUT/gaps.txt 2>&1
tail -45 $OUT/gaps.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
