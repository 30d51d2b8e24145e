#!/bin/bash
# Counters of the IDW kernels inside the bench step.  Usage: bash tools/gpu_idw_pmc.sh <tag>
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
cat > /tmp/pmc_idw.txt <<'G'
SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
G
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop"
PMC_GROUPS=/tmp/pmc_idw.txt bash tools/pmc_passes.sh $OUT/pmc $BENCH > $OUT/pmc_summary.txt 2>&1
grep -E "^idw_fine3|^idw_coarse" $OUT/pmc/summary.csv
tail -3 $OUT/pmc/p3.log
