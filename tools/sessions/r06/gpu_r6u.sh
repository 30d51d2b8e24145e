#!/bin/bash
# round 6, session u: an LK-side library variant ('new') against the tree's previous build ('base') on one box: 118 LK stage
# fingerprints, the IDW / LK suites, LK leg in alternation, per-kernel averages of the step under rocprofv3
OUT=gpurun_out/${1:-r6u}; mkdir -p $OUT; L=pysteps_amd/lib
export TMPDIR=/tmp
use() { cp $L/libpysteps_hip_$1.so $L/libpysteps_hip.so; }
{
use base; timeout 600 python tools/lk_bitcheck.py base > $OUT/bitcheck_base.log 2>&1; tail -1 $OUT/bitcheck_base.log
use new; timeout 600 python tools/lk_bitcheck.py new > $OUT/bitcheck_new.log 2>&1; tail -1 $OUT/bitcheck_new.log
python tools/lk_bitcheck.py --diff base new | tail -4
timeout 900 python -m pytest tests/test_idw_gpu.py tests/test_lk_gpu.py tests/test_lk_sparse_gpu.py -q -m gpu --timeout=400 -x -k "not 8192" 2>&1 | tail -3
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
for round in 1 2 3; do for v in base new; do use $v; echo -n "bench $v: "; timeout 300 $BENCH 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['config']['lk_ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"; done; done
for v in base new; do use $v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock > $OUT/trace_$v.log 2>&1
  echo "== kernels of the step, $v"; python tools/gap_analysis.py $OUT/trace_$v | tail -24
  find $OUT/trace_$v -name "*kernel_trace.csv" -delete; find $OUT/trace_$v -name "*agent_info.csv" -delete
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
use new
