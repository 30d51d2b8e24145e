#!/bin/bash
# Round 3, session a: parity suite + counters of the member-batched kernel (config 4 on one GPU).
# Usage (via gpurun):  bash tools/gpu_members_round.sh <tag>
set -u
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import cv2; print('cv2', cv2.__version__)" > $OUT/cv2_probe.txt 2>&1
python -c "import skimage; print('skimage', skimage.__version__)" >> $OUT/cv2_probe.txt 2>&1
free -g | head -2 > $OUT/host_mem.txt; nproc >> $OUT/host_mem.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
ENS="python tools/ensemble_quick.py 4096 6 8"
timeout 300 $ENS | tee $OUT/ensemble_quick.txt
timeout 300 python tools/ensemble_quick.py 4096 6 8 0 | tee -a $OUT/ensemble_quick.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $ENS > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/members_kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $ENS > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $ENS > $OUT/pmc_write.log 2>&1
PMC_GROUPS=tools/pmc_groups_r02.txt bash tools/pmc_passes.sh $OUT/pmc $ENS > $OUT/pmc_summary.txt 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for tag in ("fetch", "write"):
    vals = collections.defaultdict(list)
    for f in glob.glob(out + "/pmc_%s/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            if "semilag_members" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        v = v[len(v) // 2:]
        res[k + "_KB_mean_per_launch"] = sum(v) / len(v)
json.dump(res, open(out + "/members_pmc_traffic.json", "w"), indent=1)
print(res)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
