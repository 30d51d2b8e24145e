"""Time of one white-noise draw for B members on the device beside numpy's RandomState.randn
(development aid / DESIGN.md numbers).   python tools/rng_quick.py [size] [members] [draws]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pysteps_amd.device import DeviceArray, Event, synchronize
from pysteps_amd.noise.randstate import DeviceRandomStates

m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
host = [np.random.RandomState(100 + j) for j in range(B)]
hint = int(sys.argv[4]) if len(sys.argv) > 4 else T + 2
dev = DeviceRandomStates([np.random.RandomState(100 + j) for j in range(B)], m * m, n_draws=hint)
out = DeviceArray((B, m, m), np.float64)
dev.randn(m, m, out=out)
synchronize()
e0, e1 = Event(), Event()
e0.record()
for _ in range(T):
    dev.randn(m, m, out=out)
e1.record()
ms = e0.elapsed_ms(e1) / T
t = time.perf_counter()
host[0].randn(m, m)
host_s = time.perf_counter() - t
print(json.dumps({"size": m, "members": B, "n_draws_hint": hint, "device_ms_per_draw_all_members": ms, "numpy_s_per_member": host_s,
                  "values_per_s": B * m * m / ms * 1e3}))
