#!/bin/bash
# development aid: counters of the per-wave staged kernel (variant 0) and the gather kernel (variant 7)
mkdir -p gpurun_out/r03h
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT
for v in 0 7; do
  PMC_GROUPS=tools/pmc_groups_wave.txt PYSTEPS_HIP_SL_VARIANT=$v bash tools/pmc_passes.sh gpurun_out/r03h/pmc_v$v python tools/sl_quick.py 4096 24 1 sheared > gpurun_out/r03h/pmc_v$v.txt 2>&1
  grep semilag_fused gpurun_out/r03h/pmc_v$v/summary.csv
done
