#!/bin/bash
# LK: the second frame's passes beside the first frame's (PYSTEPS_HIP_LK_EARLY_PREP=1) - fingerprints and same-box A/B
mkdir -p gpurun_out/r5m
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-host-path --no-spectral --no-members-leg --no-steps-loop --no-steps-stock"
PYSTEPS_HIP_LK_EARLY_PREP=1 timeout 600 python tools/lk_bitcheck.py r5m > gpurun_out/r5m/lk_bitcheck.log 2>&1; tail -1 gpurun_out/r5m/lk_bitcheck.log
cp profiles/r04/j_lk_bitcheck_base_idw_fma.json gpurun_out/lk_bitcheck_base.json
python tools/lk_bitcheck.py --diff base r5m | tail -2
{
for round in 1 2 3; do
  echo -n "default: "; timeout 300 python bench.py --steps 20 --warmup 5 $QUICK 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['config']['lk_ms_per_step'],4))"
  echo -n "early:   "; PYSTEPS_HIP_LK_EARLY_PREP=1 timeout 300 python bench.py --steps 20 --warmup 5 $QUICK 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['config']['lk_ms_per_step'],4))"
done
} > gpurun_out/r5m/ab.txt 2>&1
cat gpurun_out/r5m/ab.txt
BENCH="python bench.py --steps 5 --warmup 2 $QUICK"
PYSTEPS_HIP_LK_EARLY_PREP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5m/trace -- $BENCH > gpurun_out/r5m/trace.log 2>&1
python tools/gap_analysis.py gpurun_out/r5m/trace > gpurun_out/r5m/gaps.txt 2>&1; cat gpurun_out/r5m/gaps.txt
find gpurun_out/r5m -name "*kernel_trace.csv" -delete; find gpurun_out/r5m -name "*agent_info.csv" -delete
