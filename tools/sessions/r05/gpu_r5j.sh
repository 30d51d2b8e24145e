#!/bin/bash
# round 5: border passes served from the window, stores with a scalar base, last-resort gathers row by row
mkdir -p gpurun_out/r5j
QUICK="--no-cpu-baseline --no-host-path --no-spectral --no-members-leg --no-steps-loop --no-steps-stock"
{
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
timeout 300 python tools/sl_bitcheck.py v0 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 v0
for f in calm inward uniform sheared; do
  echo -n "field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
done
PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 uniform 2>&1 | grep "semilag_window" | tail -1
echo -n "2048 12 K3: "; timeout 120 python tools/sl_quick.py 2048 12 3 2>&1 | tail -1
} > gpurun_out/r5j/sl.txt 2>&1
cat gpurun_out/r5j/sl.txt
timeout 300 python bench.py --steps 20 --warmup 5 $QUICK > gpurun_out/r5j/bench.json 2> gpurun_out/r5j/bench.err; cut -c1-330 gpurun_out/r5j/bench.json; echo
( timeout 900 python -m pytest tests/test_semilag_gpu.py tests/test_comm_gpu.py tests/test_callers_gpu.py tests/test_nowcast_gpu.py -q -m gpu ) > gpurun_out/r5j/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5j/pytest.txt | tail -2
( PYSTEPS_HIP_SL_VARIANT=12 timeout 600 python -m pytest tests/test_semilag_gpu.py -q -m gpu -k "not config5 and not config3" ) > gpurun_out/r5j/pytest_forced.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r5j/pytest_forced.txt | tail -2
