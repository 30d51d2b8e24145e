"""IDW variants side by side (development aid; the round-4 comparison of the fine passes in
profiles/r04/e_idw_ab.txt was made with this script when variants 2 and 3 still existed): time per call and the difference of the
fields, plus the difference from the cKDTree oracle on a sample of pixels.

    python tools/idw_ab.py [size] [L]
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from pysteps_amd import _lib
from pysteps_amd.device import Event, synchronize
from pysteps_amd.utils.interpolate import idw_to_device

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ls = [int(v) for v in sys.argv[2:]] or [900, 3600, 150]
lib = _lib.lib()
for L in Ls:
    rng = np.random.default_rng(L)
    xy = np.column_stack([rng.uniform(0, n - 1, L), rng.uniform(0, m - 1, L)])
    if L == 150:  # clustered samples: thick rings, long lists
        xy = np.clip(np.column_stack([rng.normal(n / 3, n / 40, L), rng.normal(m / 2, m / 30, L)]), 0, [n - 1, m - 1])
    uv = rng.normal(0, 2, (L, 2))
    fields = {}
    for k in (20, 5):
        for variant in (1, 0):
            _lib.check(lib.psh_set_option(b"idw_variant", variant))
            out = idw_to_device(xy, uv, m, n, k=k)
            synchronize()
            e0, e1 = Event(), Event()
            e0.record()
            for _ in range(5):
                out = idw_to_device(xy, uv, m, n, k=k)
            e1.record()
            ms = e0.elapsed_ms(e1) / 5
            fields[variant] = out.to_host()
            print("L=%d k=%d variant %d: %.3f ms/call (incl. upload)" % (L, k, variant, ms))
        d = np.abs(fields[0] - fields[1])
        print("   variants differ: max abs %.3e, fraction of pixels %.3e" % (d.max(), np.mean(d > 0)))
        # oracle on a pixel sample (float64 cKDTree, reference formula)
        from scipy.spatial import cKDTree
        ys, xs = rng.integers(0, m, 4000), rng.integers(0, n, 4000)
        dist, idx = cKDTree(xy).query(np.column_stack([xs, ys]).astype(float), k=k)
        w = 1.0 / (dist + 0.5) ** 0.5
        w /= w.sum(1, keepdims=True)
        want = np.stack([(w * uv[idx, 0]).sum(1), (w * uv[idx, 1]).sum(1)])
        for variant in (1, 0):
            got = fields[variant][:, ys, xs]
            print("   variant %d vs cKDTree sample: rel-L2 %.3e max abs %.3e" % (variant, np.linalg.norm(got - want) / np.linalg.norm(want), np.abs(got - want).max()))
_lib.check(lib.psh_set_option(b"idw_variant", 0))
