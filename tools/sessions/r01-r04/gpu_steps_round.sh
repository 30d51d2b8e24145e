#!/bin/bash
# Round 3: the real nowcasts.steps at BASELINE config 4's per-GPU size with the resident loop -
# host profile of the device run, kernel statistics of the loop.   bash tools/gpu_steps_round.sh <tag>
set -u
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/steps_quick.py 4096 6 6 --no-stock --profile > $OUT/steps_quick.json 2> $OUT/steps_host_profile.txt
tail -1 $OUT/steps_quick.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/steps_quick.py 4096 6 6 --no-stock > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/steps_kernel_stats.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
head -40 $OUT/steps_kernel_stats.csv | cut -c1-200
