#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel trace + PMC passes, idle-gap
# analysis of one step, host timeline + corner walk statistics of one estimate, FFT timings.
# Usage (from the repo root, via gpurun):  bash tools/gpu_round.sh <tag> [pytest-extra-args]
set -u
TAG=${1:-dev}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q ${2:-} 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 10 --warmup 3 2>$OUT/bench.err | tee $OUT/bench.json
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
python tools/gap_analysis.py $OUT/trace > $OUT/gaps.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
PMC_GROUPS=tools/pmc_groups_r02.txt bash tools/pmc_passes.sh $OUT/pmc $BENCH > $OUT/pmc_summary.txt 2>&1
PYSTEPS_HIP_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop 2>&1 >/dev/null | grep dense_lk | tail -4 > $OUT/lk_timeline.txt
timeout 300 python tools/fft_quick.py 2048 4096 2>&1 | tail -1 > $OUT/fft_quick.json
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
# the other BASELINE shapes (one bench line each, no CPU legs): config 2, config 5 on one GPU, config 1
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop"
{ timeout 200 python bench.py --size 2048 --frames 3 --leadtimes 12 --n-iter 3 $Q 2>/dev/null
  timeout 300 python bench.py --size 8192 --frames 2 --leadtimes 36 $Q 2>/dev/null
  timeout 200 python bench.py --size 512 --frames 2 --leadtimes 6 $Q 2>/dev/null; } > $OUT/other_shapes.jsonl
