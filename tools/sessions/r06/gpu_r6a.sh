#!/bin/bash
# round 6, first session: tile order table (border rings first) + persistent workgroups + two-rows-per-lane occupancy
# variants of the window kernel, the +-2^27 step clamp; bit check of every variant against the gather kernels, same-box
# timings in alternation, then the whole GPU suite on the new default.
OUT=gpurun_out/r6a; mkdir -p $OUT; L=pysteps_amd/lib
use() { cp $L/libpysteps_hip_$1.so $L/libpysteps_hip.so; }
{
use new
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
for cfg in "PERSIST=1" "PERSIST=0" "PERSIST=7" "PERSIST=0 RINGS=0" "WINCFG=1" "WINCFG=2" "WINCFG=1 PERSIST=5" "WINCFG=2 PERSIST=11"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $(for kv in $cfg; do echo PYSTEPS_HIP_SL_$kv; done) PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py $tag 2>&1 | tail -1
  python tools/sl_bitcheck.py --diff v7 $tag | tail -2
done
use noclamp; PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py noclamp 2>&1 | tail -1; python tools/sl_bitcheck.py --diff v7 noclamp | tail -2
t() { echo -n "$1 [$2] $3: "; env $(for kv in $2; do echo PYSTEPS_HIP_SL_$kv; done) timeout 120 python tools/sl_quick.py 4096 24 1 $3 2>&1 | tail -1 | cut -c1-62; }
for round in 1 2; do
  for f in sheared uniform; do
    use head; t head "X=0" $f
    use new; t new "PERSIST=1" $f; t new "PERSIST=0" $f; t new "PERSIST=0 RINGS=0" $f; t new "PERSIST=1 RINGS=0" $f
    t new "WINCFG=1" $f; t new "WINCFG=2" $f; t new "WINCFG=1 PERSIST=0" $f; t new "WINCFG=2 PERSIST=0" $f
    use noclamp; t noclamp "PERSIST=1" $f
  done
done
use new
PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep "semilag_window<" | tail -1
PYSTEPS_HIP_SL_STATS=1 PYSTEPS_HIP_SL_WINCFG=1 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep "semilag_window<" | tail -1
PYSTEPS_HIP_SL_STATS=1 PYSTEPS_HIP_SL_WINCFG=2 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep "semilag_window<" | tail -1
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
use new
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=400 -x ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err > $OUT/bench.json; cut -c1-400 $OUT/bench.json
