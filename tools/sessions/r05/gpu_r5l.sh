#!/bin/bash
# same-box comparison of several builds of the extrapolator (library files libpysteps_hip_<name>.so), two rounds
mkdir -p gpurun_out/r5l
L=pysteps_amd/lib
cp $L/libpysteps_hip.so $L/libpysteps_hip_new.so
{
for round in 1 2; do
  for which in ${BUILDS:-head new x1 x2 x3}; do
    cp $L/libpysteps_hip_$which.so $L/libpysteps_hip.so
    for f in uniform sheared; do
      echo -n "$which field $f: "; timeout 120 python tools/sl_quick.py 4096 24 1 $f 2>&1 | tail -1
    done
  done
done
} > gpurun_out/r5l/ab.txt 2>&1
cp $L/libpysteps_hip_new.so $L/libpysteps_hip.so
cut -c1-70 gpurun_out/r5l/ab.txt
