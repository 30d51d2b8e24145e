#!/bin/bash
# round 6, session z: with border tiles sampling from the window too - does the order of the tiles (border rings first)
# and the guard width still matter?  Environment knobs of one build, same box, three rounds
OUT=gpurun_out/${1:-r6z}; mkdir -p $OUT
export TMPDIR=/tmp
{
t() { echo -n "$1 $2: "; env $1 timeout 120 python tools/sl_quick.py 4096 24 1 $2 2>&1 | tail -1 | cut -c1-62; }
for round in 1 2 3; do for f in sheared uniform; do
  for e in X=0 PYSTEPS_HIP_SL_RINGS=0 PYSTEPS_HIP_SL_CELLS=0 PYSTEPS_HIP_SL_GUARD=1.6 PYSTEPS_HIP_SL_GUARD=2.5; do t $e $f; done
done; done
} > $OUT/knobs.txt 2>&1
cat $OUT/knobs.txt
