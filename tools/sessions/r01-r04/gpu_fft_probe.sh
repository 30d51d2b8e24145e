#!/bin/bash
# development aid: column pass of the FFT - one sweep (four_step=0) against two sweeps over tiles of 2^logt elements, 2^logc columns
mkdir -p gpurun_out/r03h
for cfg in "0:3:12" "2:3:12" "2:3:11" "2:4:11" "2:3:10"; do
  IFS=: read fs lc lt <<< "$cfg"
  echo "four_step=$fs logc=$lc logt=$lt: $(PYSTEPS_HIP_FFT_FOURSTEP=$fs PYSTEPS_HIP_FFT_STEP_LOGC=$lc PYSTEPS_HIP_FFT_STEP_LOGT=$lt python tools/fft_quick.py 4096 2048 1024 8192 2>&1 | tail -1 | python -c "import json,sys; print([(tuple(r['shape']), round(r['rfft2_ms'],4), round(r['irfft2_ms'],4), r['rel_l2_vs_numpy']) for r in json.loads(sys.stdin.read())])")"
done | tee gpurun_out/r03h/fft_probe.txt
