#!/bin/bash
# One more build of the library beside the tree's own: ONE source compiled with extra flags (or taken from a git
# revision), linked with the tree's other objects -> pysteps_amd/lib/libpysteps_hip_<name>.so (same-box A/B comparisons:
# the GPU-side script copies the variants over libpysteps_hip.so in alternation).
#   bash tools/build_variant.sh <name> <source.hip> [--rev <git rev>] [extra hipcc flags...]
set -e
NAME=$1; SRC=$2; shift 2
L=pysteps_amd/lib; O=$L/obj; B=$(basename $SRC .hip)
python -m pysteps_amd.build > /dev/null
IN=pysteps_amd/csrc/$B.hip
if [ "$1" == "--rev" ]; then
  mkdir -p /tmp/psh_variant/pysteps_amd/csrc /tmp/psh_variant/include
  cp pysteps_amd/csrc/*.h /tmp/psh_variant/pysteps_amd/csrc/; cp include/*.h /tmp/psh_variant/include/
  git show $2:pysteps_amd/csrc/$B.hip > /tmp/psh_variant/pysteps_amd/csrc/$B.hip
  for h in $(git show $2 --stat --name-only --format= -- pysteps_amd/csrc/*.h include/*.h 2>/dev/null); do :; done
  IN=/tmp/psh_variant/pysteps_amd/csrc/$B.hip; shift 2
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $IN -o /tmp/psh_variant_$NAME.o
OBJS=$(ls $O/*.o | grep -v "/$B.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libpysteps_hip_$NAME.so $OBJS /tmp/psh_variant_$NAME.o -ldl
echo $L/libpysteps_hip_$NAME.so
