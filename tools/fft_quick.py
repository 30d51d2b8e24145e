"""Timing of the HIP FFTs (development aid / docs/history.md 3.6): resident transforms by HIP events,
the NumPy-in / NumPy-out path by the wall clock, numpy.fft (pocketfft) beside them.

    python tools/fft_quick.py [size | rowsxcols ...]      e.g.  2048 4096 640x710 1226x760
"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from pysteps_amd.device import DeviceArray, Event, synchronize
from pysteps_amd.utils.fft import get_hip

sizes = sys.argv[1:] or ["2048", "4096"]
out = []
for a in sizes:
    shape = tuple(int(v) for v in a.split("x")) if "x" in a else (int(a), int(a))
    n = shape[1]
    x = np.random.default_rng(n).standard_normal(shape)
    fft = get_hip(shape)
    dx = DeviceArray.from_host(x)
    dX = fft.rfft2(dx)
    fft.irfft2(dX)
    synchronize()
    reps = 10
    e0, e1, e2 = Event(), Event(), Event()
    e0.record()
    for _ in range(reps):
        dX = fft.rfft2(dx)
    e1.record()
    for _ in range(reps):
        back = fft.irfft2(dX)
    e2.record()
    synchronize()
    fwd_ms, inv_ms = e0.elapsed_ms(e1) / reps, e1.elapsed_ms(e2) / reps
    host_fwd = host_inv = 1e9
    for _ in range(3):  # the first call also allocates the pinned result block: best of three
        t = time.perf_counter()
        X = fft.rfft2(x)
        host_fwd = min(host_fwd, time.perf_counter() - t)
        t = time.perf_counter()
        fft.irfft2(X)
        host_inv = min(host_inv, time.perf_counter() - t)
    t = time.perf_counter()
    Xn = np.fft.rfft2(x)
    np_fwd = time.perf_counter() - t
    t = time.perf_counter()
    np.fft.irfft2(Xn, s=shape)
    np_inv = time.perf_counter() - t
    err = float(np.linalg.norm(X - Xn) / np.linalg.norm(Xn))
    # compulsory traffic of a two-pass transform: real plane in, half spectrum out and once more in and out
    spec = n * (n // 2 + 1) * 16
    gb = (n * n * 8 + 3 * spec) / 1e9
    out.append({"shape": list(shape), "rfft2_ms": fwd_ms, "irfft2_ms": inv_ms, "rfft2_gbs": gb / fwd_ms * 1e3,
                "host_rfft2_ms": host_fwd * 1e3, "host_irfft2_ms": host_inv * 1e3, "numpy_rfft2_ms": np_fwd * 1e3,
                "numpy_irfft2_ms": np_inv * 1e3, "rel_l2_vs_numpy": err})
print(json.dumps(out))
