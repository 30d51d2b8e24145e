#!/bin/bash
# round 6, session g / h: lk_corner_select variants (g: 16 waves x 1 group against 4 x 4; h: the waves of a workgroup side by side against stacked):
# the committed base, same-box LK leg of the bench + kernel averages
OUT=gpurun_out/${1:-r6g}; mkdir -p $OUT; L=pysteps_amd/lib
export TMPDIR=/tmp
use() { cp $L/libpysteps_hip_$1.so $L/libpysteps_hip.so; }
use new
timeout 600 python tools/lk_bitcheck.py r6g > $OUT/bitcheck.log 2>&1; tail -1 $OUT/bitcheck.log
cp profiles/r04/j_lk_bitcheck_base_idw_fma.json gpurun_out/lk_bitcheck_base.json; python tools/lk_bitcheck.py --diff base r6g | tail -3
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
for round in 1 2 3; do for which in sel4 new; do use $which; echo -n "$which: "; timeout 300 $BENCH 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['config']['lk_ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; done; done
use new
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock > $OUT/trace.log 2>&1
python tools/gap_analysis.py $OUT/trace | tail -24
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
timeout 900 python -m pytest tests/test_lk_gpu.py tests/test_lk_sparse_gpu.py tests/test_lk_banded_gpu.py -q -m gpu --timeout=400 2>&1 | tail -3
