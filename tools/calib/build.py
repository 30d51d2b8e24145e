"""hipcc --offload-arch=gfx950 build of the tools-only calibration library."""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpsh_calib.so")


def build(force=False):
    src = os.path.join(HERE, "calib.hip")
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(src):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB_PATH, src])
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True))
