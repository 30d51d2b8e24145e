#!/bin/bash
# round 5, fourth GPU call: window kernel as the default; pipelined blends, guard width, short calls
mkdir -p gpurun_out/r5d
{
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
timeout 300 python tools/sl_bitcheck.py v0 2>&1 | tail -1
PYSTEPS_HIP_SL_PIPE=1 timeout 300 python tools/sl_bitcheck.py v0pipe 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 v0
python tools/sl_bitcheck.py --diff v7 v0pipe
for f in sheared uniform; do
  echo -n "default field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  echo -n "pipe field $f: "; PYSTEPS_HIP_SL_PIPE=1 timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
done
for g in 2.0 2.2 3.0; do
  echo -n "guard $g sheared: "; PYSTEPS_HIP_SL_GUARD=$g timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | tail -1
done
echo -n "pipe guard 2.2 sheared: "; PYSTEPS_HIP_SL_PIPE=1 PYSTEPS_HIP_SL_GUARD=2.2 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | tail -1
for T in 1 2 3 4 8; do
  for v in 12 7; do
    echo -n "T=$T variant $v: "; PYSTEPS_HIP_SL_VARIANT=$v timeout 120 python tools/sl_quick.py 4096 $T 1 2>&1 | tail -1
  done
done
} > gpurun_out/r5d/sl.txt 2>&1
cat gpurun_out/r5d/sl.txt
( time timeout 600 python -m pytest tests/test_semilag_gpu.py tests/test_comm_gpu.py tests/test_callers_gpu.py tests/test_robustness_gpu.py tests/test_nowcast_gpu.py -q -m gpu ) > gpurun_out/r5d/pytest.txt 2>&1; tail -4 gpurun_out/r5d/pytest.txt
timeout 300 python bench.py > gpurun_out/r5d/bench.json 2> gpurun_out/r5d/bench.err; cut -c1-400 gpurun_out/r5d/bench.json
PYSTEPS_HIP_SL_PIPE=1 timeout 300 python bench.py > gpurun_out/r5d/bench_pipe.json 2>> gpurun_out/r5d/bench.err; cut -c1-300 gpurun_out/r5d/bench_pipe.json
