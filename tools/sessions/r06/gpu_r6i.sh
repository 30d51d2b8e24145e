#!/bin/bash
# round 6, session i: the fused response + 3x3 maxima + compaction pass (lk_corner_response_nms) against response + select
# (PYSTEPS_HIP_LK_FUSED_NMS=0): LK fingerprints against the committed base, LK suites, same-box LK leg, kernels of a step
OUT=gpurun_out/${1:-r6i}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/lk_bitcheck.py r6i > $OUT/bitcheck.log 2>&1; tail -1 $OUT/bitcheck.log
cp profiles/r04/j_lk_bitcheck_base_idw_fma.json gpurun_out/lk_bitcheck_base.json; python tools/lk_bitcheck.py --diff base r6i | tail -6
timeout 900 python -m pytest tests/test_lk_gpu.py tests/test_lk_sparse_gpu.py -q -m gpu --timeout=400 -x 2>&1 | tail -5
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
for round in 1 2 3; do for v in 0 1; do echo -n "fused=$v: "; PYSTEPS_HIP_LK_FUSED_NMS=$v timeout 300 $BENCH 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['config']['lk_ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; done; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock > $OUT/trace.log 2>&1
python tools/gap_analysis.py $OUT/trace | tail -24
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
