"""Quick on-GPU timing of the fused semilag kernel (development aid, not the bench)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pysteps_amd.device import DeviceArray, Event, synchronize, device_info
from pysteps_amd.extrapolation import get_method
from tools import synth

def main():
    m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if len(sys.argv) > 5:  # fewer rows, same columns (keeps the planes inside the L2s)
        m = int(sys.argv[5])
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    print(device_info())
    ex = get_method("semilagrangian")
    p = DeviceArray.from_host(synth.rain_field_db(m, n))
    vel = synth.true_velocity(m, n)
    if len(sys.argv) > 4 and sys.argv[4] == "uniform":  # rigid translation: no integer crossings inside a wave
        vel[0], vel[1] = 4.3, -3.1
    if len(sys.argv) > 4 and sys.argv[4] == "calm":  # nearly at rest: no window fills, no samples leaving the image
        vel[0], vel[1] = 0.23, -0.17
    if len(sys.argv) > 4 and sys.argv[4] == "inward":  # the uniform field's speeds, every sample stays inside the image
        yy, xx = np.mgrid[0:m, 0:n]
        vel[0] = np.where(xx < n // 2, -4.3, 4.3)
        vel[1] = np.where(yy < m // 2, -3.1, 3.1)
    if len(sys.argv) > 5:
        vel *= 24.0 / T  # same total displacement as the 24-step workload
    v = DeviceArray.from_host(vel)
    if len(sys.argv) > 6 and sys.argv[6] == "noprecip":  # trajectories only: no field taps, no stores
        p = None
        kw = dict(return_displacement=True)
    else:
        kw = {}
    for _ in range(2):
        out = ex(p, v, T, outval=-15.0, n_iter=K, **kw)
    synchronize()
    reps = 5
    e0, e1 = Event(), Event()
    e0.record()
    for _ in range(reps):
        out = ex(p, v, T, outval=-15.0, n_iter=K, **kw)
    e1.record()
    ms = e0.elapsed_ms(e1) / reps
    balg = (16 * K + 8) * m * n * T
    print("semilag %dx%d T=%d K=%d: %.3f ms/call  alg %.1f GB/s (%.1f%% of 8 TB/s)  %.0f Mpx*steps/s"
          % (m, n, T, K, ms, balg / ms / 1e6, balg / ms / 1e6 / 8000 * 100, m * n * T / ms / 1e3))

main()
