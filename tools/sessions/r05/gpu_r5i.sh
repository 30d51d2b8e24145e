#!/bin/bash
# what the window kernel's border passes and fills cost: the same call on a calm field (no fills, no samples leaving the
# image), on a field with the uniform field's speeds pointing inward everywhere (fills, no border passes), uniform (both)
mkdir -p gpurun_out/r5i
{
for f in calm inward uniform sheared; do
  echo -n "field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  PYSTEPS_HIP_SL_STATS=1 timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | grep "semilag_window" | tail -1
done
} > gpurun_out/r5i/sl.txt 2>&1
cat gpurun_out/r5i/sl.txt
