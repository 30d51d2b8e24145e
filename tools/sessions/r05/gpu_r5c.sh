#!/bin/bash
# round 5, third GPU call: the planar pair-packed window kernel (variant 12)
mkdir -p gpurun_out/r5c
{
for v in 0 12; do PYSTEPS_HIP_SL_VARIANT=$v timeout 300 python tools/sl_bitcheck.py v$v 2>&1 | tail -1; done
python tools/sl_bitcheck.py --diff v0 v12
for v in 12 11; do
  for f in sheared uniform; do
    echo -n "variant $v field $f: "; PYSTEPS_HIP_SL_VARIANT=$v timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  done
done
echo -n "variant 12 2048 12 K3: "; PYSTEPS_HIP_SL_VARIANT=12 timeout 120 python tools/sl_quick.py 2048 12 3 2>&1 | tail -1
PYSTEPS_HIP_SL_STATS=1 PYSTEPS_HIP_SL_VARIANT=12 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep semilag_window | tail -1
timeout 600 python -m pytest tests/test_semilag_gpu.py -q -m gpu -k "variants or config3 or config2" 2>&1 | tail -3
} > gpurun_out/r5c/sl.txt 2>&1
cat gpurun_out/r5c/sl.txt
PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python bench.py > gpurun_out/r5c/bench_v12.json 2> gpurun_out/r5c/bench.err; cut -c1-700 gpurun_out/r5c/bench_v12.json
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT
PMC_GROUPS=tools/pmc_groups_sq2.txt PYSTEPS_HIP_SL_VARIANT=12 bash tools/pmc_passes.sh gpurun_out/r5c/pmc_v12 python tools/sl_quick.py 4096 24 1 sheared > gpurun_out/r5c/pmc_v12.txt 2>&1
grep "semilag_" gpurun_out/r5c/pmc_v12/summary.csv
rm -rf gpurun_out/r5c/pmc_v12/p*
