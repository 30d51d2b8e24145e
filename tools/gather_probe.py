"""Gather cost by access width, lane stride, alignment and residency (development aid): CU clocks
per wave64 buffer load.  Under ``rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace`` the
per-dispatch counter divided by the loads issued gives the L1 tag accesses per load of each pattern
(dispatch order = the order printed here).

    python tools/gather_probe.py [clock_hz] [short]
"""
import ctypes
import sys
sys.path.insert(0, ".")
import numpy as np
from pysteps_amd.device import DeviceArray, synchronize, device_info
from tools import calib

info = device_info()
clk = float(sys.argv[1]) if len(sys.argv) > 1 else 2.4e9
short = len(sys.argv) > 2
pitch = 16384
max_rows = 64 if short else 16384  # 256 MiB
src = DeviceArray((max_rows, pitch // 4), np.float32).fill_bytes(0)
sink = DeviceArray((256,), np.float32)
synchronize()
lib = calib.lib()
iters = 200 if short else 1000
bpc = 8
# (rows walked, width, lane stride, shift, active lanes)
# rows: 8 -> L1 resident; 1024 -> L2 of each XCD; 16384 -> MALL / HBM
cases = [(8, w, w, s, 64) for w in (1, 2, 4) for s in (0, 1)]
# overlapping footprints (the packed semi-Lagrangian gathers): b128 at 8-byte lane stride,
# b64 at 4-byte lane stride, at different alignments of lane 0
cases += [(8, 4, 2, s, 64) for s in (0, 2, 6, 10, 14)]
cases += [(8, 2, 1, s, 64) for s in (0, 1, 7, 15)]
cases += [(8, 1, 1, s, 64) for s in (8, 15)]
cases += [(8, 4, 4, s, 64) for s in (4, 8, 12)]
cases += [(8, 2, 2, s, 64) for s in (2, 8, 14)]
if not short:
    cases += [(8, w, w, 1, a) for w in (1, 4) for a in (32, 16, 4, 1)]          # exec-masked loads
    cases += [(r, w, w, 1, 64) for r in (1024, 16384) for w in (1, 4)]            # L2 / HBM resident
ms = ctypes.c_float()
for n_rows, width, stride, shift, active in cases:
    for rep in range(2):  # the second launch is the measurement
        calib.check(lib.calib_gather(src.ptr, sink.ptr, pitch, width, stride, shift, iters, bpc, n_rows, active,
                                     ctypes.byref(ms)), "calib_gather")
    loads_per_cu = bpc * 4 * iters * 8
    clks = ms.value * 1e-3 * clk / loads_per_cu
    print("rows %5d width %d stride %d shift %2d active %2d: %.3f ms  %.1f clk per wave load  (%d loads per launch)"
          % (n_rows, width, stride, shift, active, ms.value, clks, loads_per_cu * info["cu_count"]))
