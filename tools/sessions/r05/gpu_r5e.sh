#!/bin/bash
# round 5, fifth GPU call: XCD cell mapping, one barrier less per fill, guard 2.0; where the LK leg's 40 us went
mkdir -p gpurun_out/r5e
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-host-path --no-spectral --no-members-leg --no-steps-loop --no-steps-stock"
{
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
timeout 300 python tools/sl_bitcheck.py v0 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 v0
for f in sheared uniform; do
  echo -n "cells on, field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  echo -n "cells off, field $f: "; PYSTEPS_HIP_SL_CELLS=0 timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
done
echo -n "2048 12 K3: "; timeout 120 python tools/sl_quick.py 2048 12 3 2>&1 | tail -1
PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep semilag_window | tail -1
} > gpurun_out/r5e/sl.txt 2>&1
cat gpurun_out/r5e/sl.txt
timeout 300 python bench.py $QUICK > gpurun_out/r5e/bench.json 2> gpurun_out/r5e/bench.err; cut -c1-330 gpurun_out/r5e/bench.json; echo
PYSTEPS_HIP_SL_CELLS=0 timeout 300 python bench.py $QUICK > gpurun_out/r5e/bench_cells_off.json 2>> gpurun_out/r5e/bench.err; cut -c1-330 gpurun_out/r5e/bench_cells_off.json; echo
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python bench.py $QUICK > gpurun_out/r5e/bench_gathers.json 2>> gpurun_out/r5e/bench.err; cut -c1-330 gpurun_out/r5e/bench_gathers.json; echo
ROOT=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5e/trace -o t -- python $ROOT/bench.py --steps 5 --warmup 2 $QUICK > $ROOT/gpurun_out/r5e/trace_bench.json 2> $ROOT/gpurun_out/r5e/trace.log)
find gpurun_out/r5e/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -24 {} | cut -c1-60,200-330'
find gpurun_out/r5e -name "*agent_info.csv" -delete
