#!/bin/bash
# development aid: column pass of the FFT - four-step sweeps on / off, FFT tests
mkdir -p gpurun_out/r03h
for fs in 1 0; do
  echo "four_step=$fs: $(PYSTEPS_HIP_FFT_FOURSTEP=$fs python tools/fft_quick.py 4096 2048 1024 8192 4096x1024 640x710 2>&1 | tail -1 | python -c "import json,sys; print([(tuple(r['shape']), round(r['rfft2_ms'],4), round(r['irfft2_ms'],4), r['rel_l2_vs_numpy']) for r in json.loads(sys.stdin.read())])")"
done | tee gpurun_out/r03h/fft_probe.txt
timeout 600 python -m pytest tests/test_fft_gpu.py tests/test_cascade_gpu.py -x -q -m gpu 2>&1 | tail -3
