#!/bin/bash
# round 6, session e: the member update with the bit-mask kernels: per-kernel statistics, gaps of one update in launch order
OUT=gpurun_out/r6e; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-lk --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-stock"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mtrace -- $B > $OUT/mtrace.log 2>&1
python tools/gap_update.py $OUT/mtrace > $OUT/member_update_kernels.txt 2>&1
python tools/gap_member_detail.py $OUT/mtrace > $OUT/member_update_gaps.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/member_update_gaps.txt
