#!/bin/bash
# round 6, session o: the early-weights window kernel as the tree's default against the previous build (base): bit check,
# the extrapolator's suites (also with the window forced), same-box timings
OUT=gpurun_out/${1:-r6o}; mkdir -p $OUT; L=pysteps_amd/lib
export TMPDIR=/tmp
use() { cp $L/libpysteps_hip_$1.so $L/libpysteps_hip.so; }
{
use base; PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py base 2>&1 | tail -1
use new; PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py new 2>&1 | tail -1
python tools/sl_bitcheck.py --diff base new | tail -1
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 new | tail -1
timeout 900 python -m pytest tests/test_semilag_gpu.py tests/test_robustness_gpu.py -q -m gpu -x --timeout=400 2>&1 | tail -3
PYSTEPS_HIP_SL_VARIANT=12 timeout 900 python -m pytest tests/test_semilag_gpu.py -q -m gpu -x --timeout=400 -k "not config5 and not config3" 2>&1 | tail -2
t() { echo -n "$1 $2: "; timeout 120 python tools/sl_quick.py 4096 24 1 $2 2>&1 | tail -1 | cut -c1-62; }
for round in 1 2 3; do for f in sheared uniform; do for v in base new; do use $v; t $v $f; done; done; done
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
for round in 1 2 3; do for v in base new; do use $v; echo -n "bench $v: "; timeout 300 $BENCH 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['config']['lk_ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; done; done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
use new
