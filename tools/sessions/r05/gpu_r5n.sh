#!/bin/bash
# same-box comparison of builds (head, x1 = field taps by ds_read2_b32, x2 = wave-uniform all-live stores, x3 = both) + guard widths
mkdir -p gpurun_out/r5n
L=pysteps_amd/lib
{
for which in x1 x2 x3; do
  cp $L/libpysteps_hip_$which.so $L/libpysteps_hip.so
  echo -n "$which: "; timeout 300 python tools/sl_bitcheck.py $which 2>&1 | tail -1
done
cp $L/libpysteps_hip_head.so $L/libpysteps_hip.so
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
for which in x1 x2 x3; do python tools/sl_bitcheck.py --diff v7 $which; done
for round in 1 2; do
  for which in head x1 x2 x3; do
    cp $L/libpysteps_hip_$which.so $L/libpysteps_hip.so
    for f in uniform sheared; do
      echo -n "$which field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1 | cut -c1-62
    done
  done
done
cp $L/libpysteps_hip_head.so $L/libpysteps_hip.so
for g in 1.6 1.8 2.0; do echo -n "head guard $g sheared: "; PYSTEPS_HIP_SL_GUARD=$g timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | tail -1 | cut -c1-62; done
} > gpurun_out/r5n/ab.txt 2>&1
cp $L/libpysteps_hip_head.so $L/libpysteps_hip.so
cat gpurun_out/r5n/ab.txt
