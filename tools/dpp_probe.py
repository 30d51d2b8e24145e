"""Which source lane do the DPP wavefront / row shifts deliver on this device? (development aid)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from pysteps_amd.device import DeviceArray, device_info, synchronize
from tools import calib

print(device_info())
out = DeviceArray((6, 64), np.int32)
synchronize()
calib.check(calib.lib().calib_dpp(out.ptr), "calib_dpp")
res = out.to_host()
for name, row in zip(("wave_shl:1", "wave_shr:1", "wave_rol:1", "wave_ror:1", "row_shl:1", "row_shr:1"), res):
    delta = sorted(set(int(v) - i for i, v in enumerate(row) if v >= 0))
    print("%-11s source - lane in %s; lanes without data: %s" % (name, delta, [i for i, v in enumerate(row) if v < 0]))
    print("            ", row.tolist())
