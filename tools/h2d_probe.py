"""Upload rate of pageable NumPy arrays through psh_memcpy_h2d (development aid).  Round 3 tried cutting large
uploads over several host threads with their own streams and pinned staging buffers: the runtime's own staged
copy already runs at 56 GB/s on the GPU box (4 / 6 / 8 threads: 54 / 53 / 52), so that code was not kept
(profiles/r03/j_h2d_probe.txt)."""
import json
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from pysteps_amd import _lib
from pysteps_amd.device import DeviceArray

lib = _lib.lib()
out = []
for mb in (32, 268, 1610):
    a = np.random.default_rng(mb).random(mb * (1 << 20) // 8)
    d = DeviceArray(a.shape, np.float64)
    best = 1e9
    for _ in range(3):
        _lib.check(lib.psh_sync())
        t = time.perf_counter()
        _lib.check(lib.psh_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        _lib.check(lib.psh_sync())
        best = min(best, time.perf_counter() - t)
    ok = bool(np.array_equal(d.to_host(), a))
    out.append({"MiB": mb, "GBps": a.nbytes / best / 1e9, "equal": ok})
print(json.dumps({"threads": os.environ.get("PYSTEPS_HIP_H2D_THREADS", "default"), "uploads": out}))
