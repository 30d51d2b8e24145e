#!/bin/bash
# round 6, session c: saturating position arithmetic (unscaled window column + v_add_i32 clamp) against session b's build
# (+-2^27 step clamp, pre-scaled column): bit check, the sentinel-velocity tests, same-box timings; then the kernels of
# one STEPS member update (tools/gpu_member_round.sh) as the starting point of the member-update work.
OUT=gpurun_out/r6c; mkdir -p $OUT; L=pysteps_amd/lib
use() { cp $L/libpysteps_hip_$1.so $L/libpysteps_hip.so; }
{
use new
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py w12 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 w12 | tail -1
use r6b
PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py r6b 2>&1 | tail -1
python tools/sl_bitcheck.py --diff r6b w12 | tail -1
use new
timeout 600 python -m pytest tests/test_semilag_gpu.py -q -m gpu -x --timeout=300 2>&1 | tail -4
t() { echo -n "$1 $2: "; timeout 120 python tools/sl_quick.py 4096 24 1 $2 2>&1 | tail -1 | cut -c1-62; }
for round in 1 2 3; do
  for f in sheared uniform; do
    use r6b; t r6b $f
    use new; t new $f
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
use new
bash tools/gpu_member_round.sh r6c_member > $OUT/member_round.log 2>&1
tail -45 $OUT/member_round.log
