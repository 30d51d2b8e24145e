#!/bin/bash
# same-box A/B of two builds of the library (new: libpysteps_hip.so as shipped; head: libpysteps_hip_head.so), alternating
mkdir -p gpurun_out/r5k
QUICK="--no-cpu-baseline --no-host-path --no-spectral --no-members-leg --no-steps-loop --no-steps-stock"
L=pysteps_amd/lib
cp $L/libpysteps_hip.so $L/new.so.keep
{
for round in 1 2; do
  for which in new head; do
    if [ $which = new ]; then cp $L/new.so.keep $L/libpysteps_hip.so; else cp $L/libpysteps_hip_head.so $L/libpysteps_hip.so; fi
    for f in uniform sheared; do
      echo -n "$which field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
    done
    echo -n "$which bench: "; timeout 300 python bench.py --steps 20 --warmup 5 $QUICK 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['config']['lk_ms_per_step'],4))"
  done
done
} > gpurun_out/r5k/ab.txt 2>&1
cp $L/new.so.keep $L/libpysteps_hip.so
cat gpurun_out/r5k/ab.txt
