"""Run the known-traffic copies under rocprofv3 --pmc to calibrate FETCH_SIZE/WRITE_SIZE.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- python tools/calib_copy.py
Each kernel moves exactly 1 GiB in and 1 GiB out per launch (buffers > 256 MiB LLC).
"""
import sys
sys.path.insert(0, ".")
import numpy as np
from pysteps_amd.device import DeviceArray, synchronize
from tools import calib

n = 1 << 28  # 2^28 floats = 1 GiB
src = DeviceArray((n,), np.float32).fill_bytes(1)
dst = DeviceArray((n,), np.float32)
synchronize()
for vec in (1, 4):
    for _ in range(3):
        calib.check(calib.lib().calib_copy(dst.ptr, src.ptr, n, vec), "calib_copy")
print("calib done: %d bytes read + %d bytes written per launch" % (4 * n, 4 * n))
