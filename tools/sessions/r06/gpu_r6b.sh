#!/bin/bash
# round 6, session b: the window kernel with the tile order table (border rings first) as the default, persistence as a
# template instantiation behind PYSTEPS_HIP_SL_PERSIST; bit check, same-box timings, whole GPU suite, bench line,
# config-5 world-1 lines (replicated / banded motion estimate).
OUT=gpurun_out/r6b; mkdir -p $OUT; L=pysteps_amd/lib
use() { cp $L/libpysteps_hip_$1.so $L/libpysteps_hip.so; }
{
use new
PYSTEPS_HIP_SL_VARIANT=7 timeout 300 python tools/sl_bitcheck.py v7 2>&1 | tail -1
for cfg in "PERSIST=0" "PERSIST=1" "PERSIST=7" "PERSIST=0 RINGS=0" "PERSIST=0 CELLS=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $(for kv in $cfg; do echo PYSTEPS_HIP_SL_$kv; done) PYSTEPS_HIP_SL_VARIANT=12 timeout 300 python tools/sl_bitcheck.py $tag 2>&1 | tail -1
  python tools/sl_bitcheck.py --diff v7 $tag | tail -1
done
t() { echo -n "$1 [$2] $3: "; env $(for kv in $2; do echo PYSTEPS_HIP_SL_$kv; done) timeout 120 python tools/sl_quick.py 4096 24 1 $3 2>&1 | tail -1 | cut -c1-62; }
for round in 1 2 3; do
  for f in sheared uniform; do
    use head; t head "X=0" $f
    use new; t new "PERSIST=0" $f; t new "PERSIST=1" $f; t new "PERSIST=0 RINGS=0" $f
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
use new
rm -f gpurun_out/sl_seen.jsonl gpurun_out/update_flips_seen.jsonl
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=400 ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err > $OUT/bench.json; cut -c1-300 $OUT/bench.json; echo
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6b/bench.json")); print(json.dumps(d["roofline"])[:1500]); print(d["config"]["lk_ms_per_step"])
PY
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 2>$OUT/config5.err > $OUT/bench_config5_world1.json; cut -c1-400 $OUT/bench_config5_world1.json; echo
timeout 900 python bench.py --workload config5 --config5-lk banded --steps 3 --warmup 1 2>>$OUT/config5.err > $OUT/bench_config5_banded_world1.json; cut -c1-300 $OUT/bench_config5_banded_world1.json; echo
timeout 600 python bench.py --force-members-path --steps 3 --warmup 1 2>$OUT/members.err > $OUT/bench_members_world1.json; cut -c1-300 $OUT/bench_members_world1.json; echo
