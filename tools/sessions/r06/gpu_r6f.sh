#!/bin/bash
# round 6, session f: PMC counters of the motion estimate's kernels inside the bench step (where does lk_corner_select's
# time go?), the SL suite + bit check with the two-ballot window test, the assertion build over the SL / LK / IDW / CDF suites
OUT=gpurun_out/r6f; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py w12 2>&1 | tail -1
python tools/sl_bitcheck.py --diff v7 w12 | tail -1
for f in sheared uniform; do echo -n "new $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1 | cut -c1-62; done
cp pysteps_amd/lib/libpysteps_hip_r6b.so /tmp/keep.so
PMC_GROUPS=tools/pmc_groups_r02.txt bash tools/pmc_passes.sh $OUT/pmc $BENCH > $OUT/pmc_summary.txt 2>&1
grep -E "lk_corner_select|lk_corner_response|lk_track_rows|corner_order|lk_open_bits|lk_stats1|lk_to_u8" $OUT/pmc/summary.csv | head -80
bash tools/gpu_debug_build.sh r6f
