#!/bin/bash
# round 6, session d: custom xy_coords and interp_order 2 / 4 / 5 natively (new goldens from the reference), the whole
# extrapolator suite, bit check against session c's build.
OUT=gpurun_out/r6d; mkdir -p $OUT; L=pysteps_amd/lib
{
timeout 900 python -m pytest tests/test_semilag_gpu.py -q -m gpu --timeout=300 2>&1 | tail -25
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py w12 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 w12 | tail -1
cp $L/libpysteps_hip_r6b.so $L/libpysteps_hip.so
PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py r6b 2>&1 | tail -1
python tools/sl_bitcheck.py --diff r6b w12 | tail -3
cp $L/libpysteps_hip_new.so $L/libpysteps_hip.so
} > $OUT/log.txt 2>&1
cat $OUT/log.txt
