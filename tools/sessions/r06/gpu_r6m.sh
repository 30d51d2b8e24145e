#!/bin/bash
# round 6, session m: the blob feature detector (csrc/blob.hip) - goldens of the reference, SciPy's cube bit for bit, NaN
# semantics of the maximum filter, fd_method="blob" through dense_lucaskanade; timing of a 4096^2 detection
OUT=gpurun_out/${1:-r6m}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_blob_gpu.py -q -m gpu --timeout=600 2>&1 | tail -40
timeout 600 python -m pytest tests/test_lk_gpu.py -q -m gpu --timeout=400 -x -k "not 8192 and not 4096" 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | tee $OUT/blob_4096.txt
import time, numpy as np
from pysteps_amd.device import DeviceArray
from pysteps_amd.feature.blob import detection
from pysteps_amd import _lib
from tools import synth
img = synth.rain_field_db(4096, 4096, seed=3).astype(np.float64)
d = DeviceArray.from_host(img)
for method in ("log", "dog"):
    detection(d, method=method)
    _lib.lib().psh_sync()
    t0 = time.perf_counter(); pts = detection(d, method=method, return_sigmas=True); dt = time.perf_counter() - t0
    print("4096x4096 float64 blob.detection(method=%r): %d blobs, %.1f ms (10 scales 3..20, resident image)" % (method, pts.shape[0], dt * 1e3))
PY
