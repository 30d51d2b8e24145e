#!/bin/bash
# development aid: column-pass shapes of the FFT (columns per workgroup x threads), FFT tests
mkdir -p gpurun_out/r03h
for cfg in ${1:-"0:0"}; do
  c=${cfg%%:*}; t=${cfg##*:}
  echo "cols=$c threads=$t: $(PYSTEPS_HIP_FFT_COLS=$c PYSTEPS_HIP_FFT_COL_THREADS=$t python tools/fft_quick.py 4096 2048 1024 640x710 2>&1 | tail -1 | python -c "import json,sys; print([(r['shape'][0], round(r['rfft2_ms'],4), round(r['irfft2_ms'],4), r['rel_l2_vs_numpy']) for r in json.loads(sys.stdin.read())])")"
done | tee gpurun_out/r03h/fft_probe.txt
timeout 600 python -m pytest tests/test_fft_gpu.py -x -q -m gpu 2>&1 | tail -2
