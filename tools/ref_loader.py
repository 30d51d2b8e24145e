"""Load individual modules of the read-only reference (/root/reference) by path.

Test/tooling helper, only usable in the build container (the GPU box has no
/root/reference).  ``import pysteps`` itself needs jsmin/jsonschema and two
compiled Cython modules, so the package ``__init__`` files are bypassed with
stub packages whose ``__path__`` points into the reference tree; only the leaf
modules of the advection hot path (numpy/scipy-only imports) are executed.
"""

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PYSTEPS_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pysteps"))


def _stub(name, relpath):
    if name in sys.modules:
        return sys.modules[name]
    mod = types.ModuleType(name)
    mod.__path__ = [os.path.join(REFERENCE_ROOT, relpath)]
    mod.__pysteps_reference_stub__ = True
    sys.modules[name] = mod
    return mod


def load(name):
    """``load("pysteps.utils.interpolate")`` -> reference module object."""
    if not available():
        raise ImportError("reference tree not available at %s" % REFERENCE_ROOT)
    real = sys.modules.get("pysteps")
    if real is not None and not getattr(real, "__pysteps_reference_stub__", False):
        return importlib.import_module(name)  # a genuine pysteps is installed
    _stub("pysteps", "pysteps")
    for sub in ("utils", "extrapolation", "motion", "feature", "tracking"):
        _stub("pysteps." + sub, os.path.join("pysteps", sub))
    return importlib.import_module(name)
