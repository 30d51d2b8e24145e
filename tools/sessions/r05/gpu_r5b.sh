#!/bin/bash
# round 5, second GPU call: raw single-address LDS reads (variant 11) and the counters of the window kernel
mkdir -p gpurun_out/r5b
{
for v in 0 11; do PYSTEPS_HIP_SL_VARIANT=$v timeout 300 python tools/sl_bitcheck.py v$v 2>&1 | tail -1; done
python tools/sl_bitcheck.py --diff v0 v11
for v in 10 11 0; do
  for f in sheared uniform; do
    echo -n "variant $v field $f: "; PYSTEPS_HIP_SL_VARIANT=$v timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  done
done
} > gpurun_out/r5b/sl.txt 2>&1
cat gpurun_out/r5b/sl.txt
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT
for v in ${PMC_VARIANTS:-10 11 0}; do
  PMC_GROUPS=tools/pmc_groups_wave.txt PYSTEPS_HIP_SL_VARIANT=$v bash tools/pmc_passes.sh gpurun_out/r5b/pmc_v$v python tools/sl_quick.py 4096 24 1 sheared > gpurun_out/r5b/pmc_v$v.txt 2>&1
  grep "semilag_" gpurun_out/r5b/pmc_v$v/summary.csv
  rm -rf gpurun_out/r5b/pmc_v$v/p*
done
