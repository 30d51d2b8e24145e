#!/bin/bash
# probability matching on the GPU: parity tests, timings, per-kernel trace (outputs under gpurun_out/pm)
export TMPDIR=/tmp
OUT=gpurun_out/pm
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_probmatch_gpu.py -x -q > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
ROOT=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o pm -- python $ROOT/tools/probmatch_quick.py ${SIZES:-1024 4096} > $ROOT/$OUT/quick.json 2> $ROOT/$OUT/prof.log)
cat $OUT/quick.json
