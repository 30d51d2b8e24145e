#!/bin/bash
# round 5, sixth GPU call: free-running waves (no barrier per lead step)
mkdir -p gpurun_out/r5f
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-host-path --no-spectral --no-members-leg --no-steps-loop --no-steps-stock"
{
echo -n "free, small: "; PYSTEPS_HIP_SL_FREE=1 timeout 60 python tools/sl_quick.py 1024 24 1 sheared 2>&1 | tail -1
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
PYSTEPS_HIP_SL_FREE=1 timeout 300 python tools/sl_bitcheck.py v0free 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 v0free
for f in sheared uniform; do
  echo -n "barrier, field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  echo -n "free,    field $f: "; PYSTEPS_HIP_SL_FREE=1 timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
done
echo -n "free 2048 12 K3: "; PYSTEPS_HIP_SL_FREE=1 timeout 120 python tools/sl_quick.py 2048 12 3 2>&1 | tail -1
PYSTEPS_HIP_SL_FREE=1 PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep semilag_window | tail -1
} > gpurun_out/r5f/sl.txt 2>&1
cat gpurun_out/r5f/sl.txt
timeout 300 python bench.py $QUICK > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err; cut -c1-330 gpurun_out/r5f/bench.json; echo
PYSTEPS_HIP_SL_FREE=1 timeout 300 python bench.py $QUICK > gpurun_out/r5f/bench_free.json 2>> gpurun_out/r5f/bench.err; cut -c1-330 gpurun_out/r5f/bench_free.json; echo
