"""Build the plain-C oracles into oracle/_build/liboracle.so (test infrastructure).

    python -m oracle.build

No reference source is compiled: pysteps' hot path is Python over SciPy/OpenCV
(third-party, not under /root/reference), so there is no ``oracle/_ref`` for
this path ("Oracle").
"""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")
SOURCES = ["semilag_c.c"]


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    if (
        not force
        and os.path.exists(LIB)
        and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs)
    ):
        return LIB
    # no -ffast-math: NaN propagation at zero weight is part of the semantics
    cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
