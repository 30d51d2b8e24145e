"""The C-ABI library loads and exports every symbol include/pysteps_hip.h declares (no GPU)."""

import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "pysteps_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared()
    assert "psh_semilag_dev" in names and "psh_init" in names


def test_library_builds_and_exports_header_symbols():
    from pysteps_amd import build

    path = build.build()
    lib = ctypes.CDLL(path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, "declared in pysteps_hip.h but not exported: %s" % missing


def test_ctypes_signatures_cover_header():
    from pysteps_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.psh_version().startswith(b"pysteps_hip")


def test_no_gpu_fails_loudly():
    """Without a GPU the product path raises; it never computes on the CPU."""
    import numpy as np

    from pysteps_amd import _lib
    from pysteps_amd.extrapolation import get_method

    lib = _lib.load()
    if lib.psh_init(-1) == 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.HipLibraryError):
        get_method("semilagrangian")(np.ones((8, 8), np.float32), np.ones((2, 8, 8), np.float32), 1)


def test_product_never_imports_oracle():
    """pysteps_amd must not reference the oracle (test infrastructure) anywhere."""
    pkg = os.path.join(ROOT, "pysteps_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                    offenders.append(os.path.join(base, f))
    assert not offenders


def test_window_shape_rule_needs_no_device():
    """psh_semilag_window_shape is a pure function (the LK shim asks it whether a motion field also needs its
    {u, v}-pair layout): images of at least 96 x 64 pixels (since round 6 whatever the row length: rows that are not
    16-byte aligned are filled texel by texel)."""
    from pysteps_amd import _lib

    lib = _lib.load()
    assert lib.psh_semilag_window_shape(4096, 4096) == 1 and lib.psh_semilag_window_shape(64, 96) == 1
    assert lib.psh_semilag_window_shape(640, 710) == 1  # n % 4 != 0: taken too
    assert lib.psh_semilag_window_shape(63, 4096) == 0 and lib.psh_semilag_window_shape(4096, 92) == 0
