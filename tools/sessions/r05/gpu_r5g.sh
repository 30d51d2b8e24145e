#!/bin/bash
# round 5, seventh GPU call: corner chain queued before the side stream (LK), hybrid window (SL)
mkdir -p gpurun_out/r5g
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-host-path --no-spectral --no-members-leg --no-steps-loop --no-steps-stock"
timeout 600 python tools/lk_bitcheck.py r5g > gpurun_out/r5g/lk_bitcheck.log 2>&1; tail -1 gpurun_out/r5g/lk_bitcheck.log
cp profiles/r04/j_lk_bitcheck_base_idw_fma.json gpurun_out/lk_bitcheck_base.json
python tools/lk_bitcheck.py --diff base r5g | tee gpurun_out/r5g/lk_bitcheck_diff.txt | tail -3
{
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
PYSTEPS_HIP_SL_HYBRID=1 timeout 300 python tools/sl_bitcheck.py v0hy 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 v0hy
for f in sheared uniform; do
  echo -n "planar, field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  echo -n "hybrid, field $f: "; PYSTEPS_HIP_SL_HYBRID=1 timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
done
echo -n "hybrid 2048 12 K3: "; PYSTEPS_HIP_SL_HYBRID=1 timeout 120 python tools/sl_quick.py 2048 12 3 2>&1 | tail -1
} > gpurun_out/r5g/sl.txt 2>&1
cat gpurun_out/r5g/sl.txt
timeout 300 python bench.py --steps 20 --warmup 5 $QUICK > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err; cut -c1-330 gpurun_out/r5g/bench.json; echo
PYSTEPS_HIP_SL_HYBRID=1 timeout 300 python bench.py --steps 20 --warmup 5 $QUICK > gpurun_out/r5g/bench_hybrid.json 2>> gpurun_out/r5g/bench.err; cut -c1-330 gpurun_out/r5g/bench_hybrid.json; echo
BENCH="python bench.py --steps 5 --warmup 2 $QUICK"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5g/trace -- $BENCH > gpurun_out/r5g/trace.log 2>&1
python tools/gap_analysis.py gpurun_out/r5g/trace > gpurun_out/r5g/gaps.txt 2>&1; cat gpurun_out/r5g/gaps.txt
find gpurun_out/r5g -name "*kernel_trace.csv" -delete; find gpurun_out/r5g -name "*agent_info.csv" -delete
timeout 600 python -m pytest tests/test_lk_gpu.py tests/test_lk_sparse_gpu.py tests/test_lk_banded_gpu.py -q -m gpu -x -k "not 8192 and not 4096" 2>&1 | tail -2
