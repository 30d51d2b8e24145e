#!/bin/bash
# Round 3: bench lines of every workload on one GPU.   bash tools/gpu_bench_round.sh <tag>
set -u
TAG=${1:-r03d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 3 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-600
timeout 600 python bench.py --force-members-path --steps 3 --warmup 1 2>$OUT/members.err | tee $OUT/bench_members_world1.json | cut -c1-400
timeout 600 python bench.py --force-members-path --advection-only --steps 3 --warmup 1 2>>$OUT/members.err | tee $OUT/bench_members_advection_world1.json | cut -c1-300
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 2>$OUT/config5.err | tee $OUT/bench_config5_world1.json | cut -c1-400
timeout 600 python -m pytest tests/test_comm_gpu.py -x -q 2>&1 | tail -3
for f in $OUT/*.err; do tail -n 5 $f; done
