#!/bin/bash
# Closing session of a round: the whole GPU suite, the bench line, the probability-matching timings with
# their kernel trace, the end-to-end nowcasts.steps comparison.  Usage (via gpurun, repo root):  bash tools/gpu_final_round.sh <tag>
set -u
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
echo "rc=$?" >> $OUT/pytest_gpu.txt
tail -4 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 10 --warmup 3 2>$OUT/bench.err | tee $OUT/bench.json
ROOT=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/pm_prof -o pm -- python $ROOT/tools/probmatch_quick.py 1024 4096 > $ROOT/$OUT/probmatch_quick.json 2> $ROOT/$OUT/pm_prof.log)
cat $OUT/probmatch_quick.json
python tools/probmatch_trace.py $OUT/pm_prof/pm_kernel_trace.csv 22 31 > $OUT/probmatch_calls.txt 2>&1
# the real nowcasts.steps with the stock operators and with every device piece on (wall clock, host profile)
STEPS_PROFILE=1 timeout 200 python tools/steps_quick.py 1024 4 3 > $OUT/steps_quick.json 2> $OUT/steps_host_profile.txt
cat $OUT/steps_quick.json
find $OUT -name "*agent_info.csv" -delete
