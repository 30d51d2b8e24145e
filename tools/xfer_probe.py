"""Host <-> device transfer rates behind the host-buffer path (development aid)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from pysteps_amd import _lib, _pinned
from pysteps_amd.device import DeviceArray, synchronize
from pysteps_amd.extrapolation import get_method

lib = _lib.lib()
n = 32 << 20  # floats: 128 MiB
page = np.ones(n, np.float32)
pin = _pinned.empty((n,), np.float32); pin[...] = 1.0
dev = DeviceArray((n,), np.float32)
def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        synchronize(); t0 = time.perf_counter(); fn(); synchronize(); best = min(best, time.perf_counter() - t0)
    return best
gb = page.nbytes / 1e9
print("H2D pageable hipMemcpyAsync : %.1f GB/s" % (gb / t(lambda: lib.psh_memcpy_h2d(dev.ptr, page.ctypes.data, page.nbytes))))
print("H2D pinned                  : %.1f GB/s" % (gb / t(lambda: lib.psh_memcpy_h2d(dev.ptr, pin.ctypes.data, pin.nbytes))))
print("D2H pinned                  : %.1f GB/s" % (gb / t(lambda: lib.psh_memcpy_d2h(pin.ctypes.data, dev.ptr, pin.nbytes))))
print("D2H pageable                : %.1f GB/s" % (gb / t(lambda: lib.psh_memcpy_d2h(page.ctypes.data, dev.ptr, page.nbytes))))
print("numpy copy pageable->pinned : %.1f GB/s" % (gb / t(lambda: np.copyto(pin, page))))
big = DeviceArray((12, n), np.float32)
bigpin = _pinned.empty((12, n), np.float32)
print("D2H pinned 1.5 GiB          : %.1f GB/s" % (12 * gb / t(lambda: lib.psh_memcpy_d2h(bigpin.ctypes.data, big.ptr, bigpin.nbytes), 2)))
del big, bigpin
from tools import synth
m = 4096
p = synth.rain_field_db(m, m); v = synth.true_velocity(m, m)
ex = get_method("semilagrangian")
for T in (1, 24):
    ex(p, v, T, outval=-15.0)
    print("host call T=%2d pageable inputs: %.1f ms" % (T, 1e3 * t(lambda: ex(p, v, T, outval=-15.0))))
    pp = _pinned.empty(p.shape, np.float32); pp[...] = p
    vv = _pinned.empty(v.shape, np.float32); vv[...] = v
    print("host call T=%2d pinned inputs  : %.1f ms" % (T, 1e3 * t(lambda: ex(pp, vv, T, outval=-15.0))))
