#!/bin/bash
# round 6, session n: semilag_window with the bilinear weights computed while the LDS reads are in flight
# (-DPSH_WIN_EARLY_WEIGHTS) against the tree's build: bit check, same-box timings in alternation, bench leg
OUT=gpurun_out/${1:-r6n}; mkdir -p $OUT; L=pysteps_amd/lib
export TMPDIR=/tmp
use() { cp $L/libpysteps_hip_$1.so $L/libpysteps_hip.so; }
{
use base; PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py base 2>&1 | tail -1
for v in ${VARIANTS:-ew}; do
  use $v; PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py $v 2>&1 | tail -1
  python tools/sl_bitcheck.py --diff base $v | tail -1
done
t() { echo -n "$1 $2: "; timeout 120 python tools/sl_quick.py 4096 24 1 $2 2>&1 | tail -1 | cut -c1-62; }
for round in 1 2 3; do
  for f in sheared uniform; do
    for v in base ${VARIANTS:-ew}; do use $v; t $v $f; done
  done
done
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
for round in 1 2; do for v in base ${VARIANTS:-ew}; do use $v; echo -n "bench $v: "; timeout 300 $BENCH 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['config']['lk_ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; done; done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
use base
