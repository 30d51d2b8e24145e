"""Config-4 surrogate on one GPU: B members x 4096^2, T single-step stateful calls with BPS
perturbed velocities, everything resident (development aid / DESIGN.md numbers)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pysteps_amd.device import DeviceArray, Event, synchronize
from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
from tools import synth

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
T = int(sys.argv[3]) if len(sys.argv) > 3 else 24
pert = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
base = synth.rain_field_db(m, n)
members = DeviceArray((B, m, n), np.float32)
from pysteps_amd import _lib
for j in range(B):
    _lib.check(_lib.lib().psh_memcpy_h2d(members.view(j).ptr, base.ctypes.data, base.nbytes))
synchronize()
rng = np.random.default_rng(0)
perts = [dict(eps_par=rng.laplace(scale=0.7), eps_perp=rng.laplace(scale=0.7), p_par=(10.88, 0.23, -7.68),
              p_perp=(5.76, 0.31, -2.72), vsf=12.0) for _ in range(B)] if pert else None
packed = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
if len(sys.argv) > 6:
    from pysteps_amd import _lib as _l
    _l.check(_l.lib().psh_set_option(b"members_variant", int(sys.argv[6])))
adv = EnsembleAdvector(synth.true_velocity(m, n), B, perts, outval=-15.0, packed=packed)
cur = members
for t in range(2):
    cur = adv.step(cur, 1.0, 5.0 * (t + 1))
synchronize()
e0, e1 = Event(), Event()
e0.record()
for t in range(T):
    cur = adv.step(cur, 1.0, 5.0 * (t + 3))
e1.record()
ms = e0.elapsed_ms(e1)
px_steps = B * m * n * T
print("ensemble %dx%d B=%d T=%d pert=%s: %.2f ms total, %.3f ms/step-all-members, %.0f Mpx*leadsteps/s, %.0f GB/s (48 B/px/step)"
      % (m, n, B, T, pert, ms, ms / T, px_steps / ms / 1e3, px_steps * 48 / ms / 1e6))
