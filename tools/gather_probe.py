"""Gather throughput by access width, alignment and residency (development aid):
CU clocks per wave64 buffer load, bytes per CU clock."""
import sys
sys.path.insert(0, ".")
import numpy as np
from pysteps_amd import _lib
from pysteps_amd.device import DeviceArray, Event, synchronize, device_info

info = device_info()
clk = float(sys.argv[1]) if len(sys.argv) > 1 else 2.4e9
pitch = 16384
max_rows = 16384  # 256 MiB
src = DeviceArray((max_rows, pitch // 4), np.float32).fill_bytes(0)
sink = DeviceArray((256,), np.float32)
lib = _lib.lib()
iters = 1000
bpc = 8
# rows walked: 8 -> 2-8 KiB per wave set (L1); 64 -> 64-256 KiB (L2, shared by all CUs);
# 1024 -> 1-4 MiB (L2 of each XCD); 16384 -> 16-64 MiB window (MALL / HBM)
cases = [(8, w, s, 64) for w in (1, 2, 4) for s in (0, 1)]
cases += [(8, w, 1, a) for w in (1, 4) for a in (32, 16, 4, 1)]          # exec-masked loads
cases += [(r, w, 1, 64) for r in (1024, 16384) for w in (1, 4)]            # L2 / HBM resident
for n_rows, width, shift, active in cases:
    if True:
        if True:
            for _ in range(2):
                _lib.check(lib.psh_calib_gather(src.ptr, sink.ptr, pitch, width, shift, iters, bpc, n_rows, active))
            synchronize()
            e0, e1 = Event(), Event()
            e0.record()
            _lib.check(lib.psh_calib_gather(src.ptr, sink.ptr, pitch, width, shift, iters, bpc, n_rows, active))
            e1.record()
            ms = e0.elapsed_ms(e1)
            loads_per_cu = bpc * 4 * iters * 8
            clks = ms * 1e-3 * clk / loads_per_cu
            tbs = loads_per_cu * info["cu_count"] * 256.0 * width / (ms * 1e-3) / 1e12
            print("rows %5d width %d shift %d active lanes %2d: %.3f ms  %.1f clk per wave load  %.1f B/clk/CU  %.2f TB/s"
                  % (n_rows, width, shift, active, ms, clks, 4.0 * active * width / clks, tbs * active / 64.0))
