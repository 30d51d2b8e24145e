#!/bin/bash
# One GPU-box session: parity suites of the pieces that changed, semilag variant timings + bit
# fingerprints, FFT timings, a bench line without the CPU legs, kernel trace + idle-gap analysis.
# Usage (via gpurun):  bash tools/gpu_mixed_round.sh <tag> "<semilag variants>"
set -u
TAG=${1:-mixed}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_lk_sparse_gpu.py tests/test_fft_gpu.py tests/test_semilag_gpu.py \
  tests/test_ensemble_gpu.py tests/test_nowcast_gpu.py tests/test_lk_gpu.py -q 2>&1 | tail -60 > $OUT/pytest.txt
tail -12 $OUT/pytest.txt
bash tools/gpu_sl_round.sh "${2:-0 5}" "${2:-0 5} 1" 2>&1 | tee $OUT/sl_variants.txt
timeout 300 python tools/fft_quick.py 2048 4096 2>&1 | tail -3 | tee $OUT/fft_quick.json
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral"
timeout 300 $BENCH 2>$OUT/bench.err | tee $OUT/bench_quick.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
python tools/gap_analysis.py $OUT/trace > $OUT/gaps.txt 2>&1
tail -40 $OUT/gaps.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
