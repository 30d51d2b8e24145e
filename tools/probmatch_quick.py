"""Timing of the device probability matching (development aid / docs/history.md 3.8): resident calls by HIP
events, the NumPy-in / NumPy-out path and the oracle (two host sorts, like the reference) by the wall
clock; the result is checked against the oracle.

    python tools/probmatch_quick.py [size ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.ndimage import gaussian_filter

from oracle import probmatch as oracle
from pysteps_amd.device import DeviceArray, Event, synchronize
from pysteps_amd.postprocessing.probmatching import nonparam_match_empirical_cdf


def field(shape, seed, wet):
    rng = np.random.default_rng(seed)
    g = gaussian_filter(rng.standard_normal(shape), 6.0, mode="wrap")
    g = 12.0 * g / g.std() + rng.normal(0, 1e-2, shape)
    out = g.copy()
    out[g < np.quantile(g, 1.0 - wet)] = -15.0
    return out


only = os.environ.get("PM_CASE")  # "masked" / "all_wet": just that case (counter passes)
sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096]
out = []
for n in sizes:
    shape = (n, n)
    for label, wet_i, wet_t in (("masked", 0.25, 0.35), ("all_wet", 1.0, 0.35)):
        if only and label != only:
            continue
        initial = field(shape, 1, wet_i)
        target = np.round(field(shape, 2, wet_t), 1)
        di, dt = DeviceArray.from_host(initial), DeviceArray.from_host(target)
        got = nonparam_match_empirical_cdf(di, dt)
        synchronize()
        reps = 5
        e0, e1 = Event(), Event()
        e0.record()
        for _ in range(reps):
            got = nonparam_match_empirical_cdf(di, dt)
        e1.record()
        synchronize()
        dev_ms = e0.elapsed_ms(e1) / reps
        host = 1e9
        for _ in range(3):
            t = time.perf_counter()
            h = nonparam_match_empirical_cdf(initial, target)
            host = min(host, time.perf_counter() - t)
        t = time.perf_counter()
        want = oracle.nonparam_match_empirical_cdf(initial, target)
        cpu = time.perf_counter() - t
        out.append({"shape": list(shape), "case": label, "resident_ms": dev_ms, "host_path_ms": host * 1e3,
                    "oracle_cpu_ms": cpu * 1e3, "bit_exact": bool(np.array_equal(got.to_host(), want) and np.array_equal(h, want))})
print(json.dumps(out))
