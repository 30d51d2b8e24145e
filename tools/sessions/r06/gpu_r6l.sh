#!/bin/bash
# round 6, session l: member update with the mask applied by the matching's first sweep (psh_steps_mask_probmatch_dev), pm_init /
# pm2_scan_sums / pm_threshold / pm2_rank_large folded into their neighbours, the mask step without its copy and clearing launch:
# the suites of the update's stages, then the steps_loop leg and the kernels of one update
OUT=gpurun_out/${1:-r6l}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_probmatch_gpu.py tests/test_masks_gpu.py tests/test_steps_resident_gpu.py tests/test_callers_gpu.py tests/test_nowcast_gpu.py -q -m gpu --timeout=400 -x 2>&1 | tail -5
B="python bench.py --steps 2 --warmup 1 --no-lk --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-stock"
for i in 1 2 3; do timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config']['steps_loop']; print('ms_per_member_update', round(s['ms_per_member_update'],4), 'ms_per_leadtime_all_members', round(s['ms_per_leadtime_all_members'],3))"; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mtrace -- $B > $OUT/mtrace.log 2>&1
python tools/gap_update.py $OUT/mtrace > $OUT/member_update_kernels.txt 2>&1; cat $OUT/member_update_kernels.txt | head -40
python tools/gap_member_detail.py $OUT/mtrace 2>&1 | tail -40
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
