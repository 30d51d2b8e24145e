#!/bin/bash
# round 5, last GPU call: counters of the final window kernel, smoke()
mkdir -p gpurun_out/r5h
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT
PMC_GROUPS=tools/pmc_groups_sq2.txt bash tools/pmc_passes.sh gpurun_out/r5h/pmc python tools/sl_quick.py 4096 24 1 sheared > gpurun_out/r5h/pmc.txt 2>&1
grep "semilag_" gpurun_out/r5h/pmc/summary.csv
rm -rf gpurun_out/r5h/pmc/p*
PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 sheared 2>&1 | grep -i "semilag" | tail -2
