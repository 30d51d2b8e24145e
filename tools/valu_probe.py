"""Issue cost of single VALU instructions on the device (development aid): cycles per wave64
instruction and SIMD at 1 / 2 / 4 / 8 waves per SIMD, from tools/calib's calib_valu kernels.

    python tools/valu_probe.py  > gpurun_out/valu_probe.txt
"""
import ctypes
import sys

sys.path.insert(0, ".")
import numpy as np

from pysteps_amd.device import DeviceArray, device_info, synchronize
from tools import calib

NAMES = ["v_fma_f32", "v_pk_fma_f32", "v_sqrt_f32", "v_rsq_f32", "v_add_f64", "v_cvt_f64_f32", "v_fma_f64",
         "v_mov_b32_dpp", "v_rcp_f32", "v_cvt_f32_f64", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f32_dpp", "v_mul_f64",
         "v_add_u32", "ds_bpermute(shfl_xor)"]


def main():
    sink = DeviceArray((1024,), np.float32)
    synchronize()
    iters = 20000
    clock_ghz = 2.4  # nominal; the ratio between rows is what matters
    print("cycles per wave64 instruction and SIMD at the nominal %.1f GHz (8 independent chains per wave)" % clock_ghz)
    print("%-24s %8s %8s %8s %8s" % ("instruction", "1 w/SIMD", "2", "4", "8"))
    for op, name in enumerate(NAMES):
        row = []
        for waves in (1, 2, 4, 8):
            ms = ctypes.c_float(0)
            for _ in range(2):
                calib.check(calib.lib().calib_valu(sink.ptr, op, waves, iters, ctypes.byref(ms)), "calib_valu")
            instr_per_simd = waves * iters * 8
            row.append(ms.value * 1e-3 * clock_ghz * 1e9 / instr_per_simd)
        print("%-24s %8.2f %8.2f %8.2f %8.2f" % ((name,) + tuple(row)))


if __name__ == "__main__":
    main()
