#!/bin/bash
# development aid: bit fingerprints + timings of the semilag kernel variants (run through gpurun)
# usage: bash tools/gpu_sl_round.sh "<variants>" [bitcheck-variants]
mkdir -p gpurun_out
for v in ${2:-}; do PYSTEPS_HIP_SL_VARIANT=$v python tools/sl_bitcheck.py v$v 2>&1 | tail -1; done
first=""
for v in ${2:-}; do if [ -z "$first" ]; then first=$v; else python tools/sl_bitcheck.py --diff v$first v$v; fi; done
for v in $1; do
  for f in sheared uniform; do
    echo -n "variant $v field $f: "; PYSTEPS_HIP_SL_VARIANT=$v python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
  done
  echo -n "variant $v 2048 12 K3: "; PYSTEPS_HIP_SL_VARIANT=$v python tools/sl_quick.py 2048 12 3 2>&1 | tail -1
done
