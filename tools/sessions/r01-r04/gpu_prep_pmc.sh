#!/bin/bash
# Counters of the frame passes and pyramid kernels inside the bench step.  Usage: bash tools/gpu_prep_pmc.sh <tag>
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop"
PMC_GROUPS=tools/pmc_groups_prep.txt bash tools/pmc_passes.sh $OUT/pmc $BENCH > $OUT/pmc_summary.txt 2>&1
grep -E "lk_stats1|lk_open_bits|lk_to_u8|lk_pyrdown|lk_corner_select|corner_order" $OUT/pmc/summary.csv
