"""Quick on-GPU timing of the IDW kernel (development aid)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from pysteps_amd.device import Event, synchronize
from pysteps_amd.utils.interpolate import idw_to_device

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(0)
xy = np.column_stack([rng.integers(0, n, L), rng.integers(0, m, L)]).astype(float)
uv = rng.normal(0, 2, (L, 2))
out = idw_to_device(xy, uv, m, n)
synchronize()
e0, e1 = Event(), Event()
reps = 3
e0.record()
for _ in range(reps):
    out = idw_to_device(xy, uv, m, n)
e1.record()
print("idw %dx%d L=%d k=20: %.3f ms/call (incl. upload+sync)" % (m, n, L, e0.elapsed_ms(e1) / reps))
