#!/bin/bash
# round 5, closing session: the whole GPU suite, the bench line with the driver's flags, rocprofv3 kernel trace + PMC
# passes of the bench step, idle gaps, LK timeline, the other BASELINE shapes, the world-1 lines of the N > 1 paths,
# the OpenCV probe.   bash tools/gpu_r5z.sh <tag>
set -u
TAG=${1:-r5z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f gpurun_out/sl_seen.jsonl gpurun_out/update_flips_seen.jsonl
( time timeout 1200 python -m pytest tests -m gpu -q --timeout=400 --durations=12 ) > $OUT/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest_gpu.txt | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err > $OUT/bench.json; cut -c1-420 $OUT/bench.json; echo
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
python tools/gap_analysis.py $OUT/trace > $OUT/gaps.txt 2>&1; tail -1 $OUT/gaps.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
PMC_GROUPS=tools/pmc_groups_r02.txt bash tools/pmc_passes.sh $OUT/pmc $BENCH > $OUT/pmc_summary.txt 2>&1
PYSTEPS_HIP_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock 2>&1 >/dev/null | grep dense_lk | tail -4 > $OUT/lk_timeline.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-host-path --no-members-leg --no-spectral --no-steps-loop --no-steps-stock"
{ timeout 200 python bench.py --size 2048 --frames 3 --leadtimes 12 --n-iter 3 $Q 2>/dev/null
  timeout 300 python bench.py --size 8192 --frames 2 --leadtimes 36 $Q 2>/dev/null
  timeout 200 python bench.py --size 512 --frames 2 --leadtimes 6 $Q 2>/dev/null; } > $OUT/other_shapes.jsonl
cut -c1-200 $OUT/other_shapes.jsonl
timeout 600 python bench.py --force-members-path --steps 3 --warmup 1 2>$OUT/members.err > $OUT/bench_members_world1.json; cut -c1-300 $OUT/bench_members_world1.json; echo
timeout 600 python bench.py --force-members-path --advection-only --steps 3 --warmup 1 2>>$OUT/members.err > $OUT/bench_members_advection_world1.json
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 2>$OUT/config5.err > $OUT/bench_config5_world1.json; cut -c1-300 $OUT/bench_config5_world1.json; echo
{ echo "== python3"; python -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1
  echo "== /opt/conda/bin/python3.9"; /opt/conda/bin/python3.9 -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1
  echo "== files named cv2* / opencv* outside /proc"; find / -xdev \( -iname "cv2*" -o -iname "*opencv*" \) -not -path "/proc/*" 2>/dev/null | head -20
  echo "== pip"; python -m pip list 2>/dev/null | grep -i -E "opencv|cv2" || echo "(no opencv package)"
  echo "== /opt/conda scikit-image"; /opt/conda/bin/python3.9 -c "import skimage; print(skimage.__version__)" 2>&1 | tail -1; } > $OUT/cv2_probe.txt 2>&1
du -sh $OUT
# the extrapolator's tests once more with the window kernel forced onto every eligible call (single-step calls too)
( PYSTEPS_HIP_SL_VARIANT=12 timeout 600 python -m pytest tests/test_semilag_gpu.py tests/test_callers_gpu.py tests/test_nowcast_gpu.py -q -m gpu -k "not config5 and not config3" ) > $OUT/pytest_window_forced.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest_window_forced.txt | tail -2
