"""Stage timing of the HIP dense LK at full size (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pysteps_amd.device import DeviceArray, synchronize
from pysteps_amd.motion import lucaskanade as lk
from pysteps_amd.utils.cleansing import decluster, detect_outliers
from pysteps_amd.utils.interpolate import idw_to_device
from pysteps_amd import extrapolation
from tools import synth

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = synth.rain_field_db(m, n)
vel = DeviceArray.from_host(synth.true_velocity(m, n))
frames = DeviceArray((2, m, n), np.float32)
from pysteps_amd import _lib
_lib.check(_lib.lib().psh_memcpy_h2d(frames.ptr, base.ctypes.data, base.nbytes))
adv = extrapolation.get_method("semilagrangian")(frames.view(0), vel, 1, outval=-15.0)
_lib.check(_lib.lib().psh_memcpy_d2d(frames.view(1).ptr, adv.ptr, adv.nbytes))
synchronize()

def T(label, fn):
    synchronize(); t0 = time.perf_counter(); r = fn(); synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("  %-28s %8.3f ms" % (label, dt)); return r

for rep in range(2):
    print("rep", rep)
    t_all = time.perf_counter()
    p0 = T("prepare frame0 (+feat)", lambda: lk.PreparedFrame(frames.view(0), 3, 5, True))
    p1 = T("prepare frame1", lambda: lk.PreparedFrame(frames.view(1), 3, 5, False))
    pts = T("corners", lambda: lk.detect_corners(p0))
    res = T("track %d pts" % len(pts), lambda: lk.track_points(p0, p1, pts))
    nxt, st = res
    xy, uv = pts[st].astype(float), (nxt - pts)[st].astype(float)
    out = T("detect_outliers (host)", lambda: detect_outliers(uv, 3, xy, 30))
    xy, uv = xy[~out], uv[~out]
    dxy, duv = T("decluster (host)", lambda: decluster(xy, uv, 20, 1))
    V = T("idw %d vectors" % len(dxy), lambda: idw_to_device(dxy, duv, m, n))
    print("  total %.3f ms; vectors %d; mean uv %s" % ((time.perf_counter() - t_all) * 1e3, len(dxy), duv.mean(0)))
V2 = T("dense_lucaskanade()", lambda: lk.dense_lucaskanade(frames))
V2 = T("dense_lucaskanade()", lambda: lk.dense_lucaskanade(frames))
