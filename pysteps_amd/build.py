"""Build libpysteps_hip.so in-tree with hipcc for gfx950.

    python -m pysteps_amd.build [--force] [--verbose] [--debug]

--debug builds pysteps_amd/lib/libpysteps_hip_debug.so beside the product (-O1 -g -DPSH_DEBUG: the device-side
assertions of common.h PSH_DASSERT - window and tile bounds of the extrapolator's window kernel, list / segment bounds
of the interpolation and probability-matching kernels - are compiled in and trap the kernel that violates one);
PYSTEPS_HIP_LIB=<path> makes the package load it (tools/gpu_debug_build.sh runs the SL + LK suites under it once per
round, the log is kept under profiles/).

The shared object lands in pysteps_amd/lib/ (git-ignored, but it travels to the
GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""

import glob
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libpysteps_hip.so")
ARCH = "gfx950"
HEADER = os.path.join(os.path.dirname(PKG_DIR), "include", "pysteps_hip.h")


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build pysteps_amd")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(SRC_DIR, "*.hip")))


def _deps():
    return glob.glob(os.path.join(SRC_DIR, "*.h")) + [HEADER]


DEBUG_LIB_PATH = os.path.join(LIB_DIR, "libpysteps_hip_debug.so")


def build(force=False, verbose=False, debug=False):
    obj_dir = os.path.join(LIB_DIR, "obj_debug") if debug else OBJ_DIR
    lib_path = DEBUG_LIB_PATH if debug else LIB_PATH
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    dep_mtime = max(os.path.getmtime(p) for p in _deps())
    objs, rebuilt = [], False
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    if debug:
        flags = ["--offload-arch=" + ARCH, "-O1", "-g", "-DPSH_DEBUG", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (
            not force
            and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(src), dep_mtime)
        ):
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        rebuilt = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if rebuilt or not os.path.exists(lib_path):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, debug="--debug" in sys.argv))
