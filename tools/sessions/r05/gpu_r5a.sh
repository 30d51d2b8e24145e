#!/bin/bash
# round 5, first GPU call: the whole GPU suite (with the new full-size parity cases), then the window kernels
# (variants 9 / 10) against the default: bit fingerprints, timings, window statistics, and the bench line
mkdir -p gpurun_out
rm -f gpurun_out/sl_seen.jsonl gpurun_out/update_flips_seen.jsonl
( time timeout 900 python -m pytest tests -m gpu -q --timeout=400 --durations=25 ) > gpurun_out/r5a_pytest.txt 2>&1
tail -5 gpurun_out/r5a_pytest.txt
{
for v in 0 9 10; do PYSTEPS_HIP_SL_VARIANT=$v timeout 300 python tools/sl_bitcheck.py v$v 2>&1 | tail -1; done
python tools/sl_bitcheck.py --diff v0 v9
python tools/sl_bitcheck.py --diff v0 v10
for v in 0 9 10; do
  for f in sheared uniform; do
    echo -n "variant $v field $f: "; PYSTEPS_HIP_SL_VARIANT=$v timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  done
  echo -n "variant $v 2048 12 K3: "; PYSTEPS_HIP_SL_VARIANT=$v timeout 120 python tools/sl_quick.py 2048 12 3 2>&1 | tail -1
done
for v in 9 10; do
  echo "stats variant $v:"; PYSTEPS_HIP_SL_STATS=1 PYSTEPS_HIP_SL_VARIANT=$v timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep semilag_window | tail -2
done
} > gpurun_out/r5a_sl.txt 2>&1
cat gpurun_out/r5a_sl.txt
timeout 300 python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; cat gpurun_out/r5a_bench.json | cut -c1-600
PYSTEPS_HIP_SL_VARIANT=10 timeout 300 python bench.py > gpurun_out/r5a_bench_v10.json 2>> gpurun_out/r5a_bench.err; cut -c1-300 gpurun_out/r5a_bench_v10.json
