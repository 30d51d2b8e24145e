#!/bin/bash
# Same-box A/B of one environment switch on the bench step: three alternating runs per value, so that the box's
# sustained clock (2-3 % between boxes) cancels.  Prints lk_ms_per_step, ms_per_step and the value per run.
# Usage (one gpurun call): bash tools/gpu_env_ab.sh <ENV_VAR> "<value_a> <value_b> ..."   ("-" = unset)
set -u
VAR=$1; VALS=$2
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop"
for rep in 1 2 3; do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', 'lk_ms', round(d['config']['lk_ms_per_step'],4), 'step_ms', round(d['ms_per_step'],4), 'value', round(d['value']))"
  done
done
