#!/bin/bash
# development aid: the per-wave staged semilag kernels against the gather kernel (variant 7):
# bit fingerprints, the semilag GPU tests, timings.  usage: gpu_sl_wave_round.sh "<variants to time>" "<bitcheck variants>" [test variant]
mkdir -p gpurun_out/r03h
( bash tools/gpu_sl_round.sh "$1" "$2" ) > gpurun_out/r03h/sl_round.txt 2>&1
if [ -n "$3" ]; then
  PYSTEPS_HIP_SL_VARIANT=$3 timeout 900 python -m pytest tests/test_semilag_gpu.py -x -q -m gpu > gpurun_out/r03h/pytest_sl.txt 2>&1
  tail -3 gpurun_out/r03h/pytest_sl.txt
fi
cat gpurun_out/r03h/sl_round.txt
