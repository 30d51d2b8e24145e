#!/bin/bash
# The extrapolator, motion-estimate, interpolation and probability-matching GPU suites once under the assertion build
# (python -m pysteps_amd.build --debug: -O1 -g -DPSH_DEBUG, PSH_DASSERT compiled in) -> gpurun_out/<tag>/debug_build.txt
#   bash tools/gpu_debug_build.sh <tag>
OUT=gpurun_out/${1:-debug}; mkdir -p $OUT
ls -la pysteps_amd/lib/libpysteps_hip_debug.so > $OUT/debug_build.txt
( time PYSTEPS_HIP_LIB=$PWD/pysteps_amd/lib/libpysteps_hip_debug.so timeout 2400 python -m pytest tests/test_semilag_gpu.py tests/test_lk_gpu.py \
    tests/test_idw_gpu.py tests/test_probmatch_gpu.py tests/test_lk_sparse_gpu.py tests/test_blob_gpu.py tests/test_masks_gpu.py -q -m gpu --timeout=900 -k "not 8192" ) >> $OUT/debug_build.txt 2>&1
tail -6 $OUT/debug_build.txt
