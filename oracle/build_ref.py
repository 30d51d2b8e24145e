"""Build ``oracle/_ref``: an importable copy of the REAL reference package (test infrastructure).

    python -m oracle.build_ref [--force]

Recipe = SURVEY.md section 8(c) "Whole-package oracle":

1. copy ``/root/reference/pysteps`` (read-only there) to the git-ignored ``oracle/_ref/pysteps``
   (never committed: ``oracle/_ref/`` is in ``.gitignore``; it travels to the GPU box with the
   gpurun snapshot exactly like the built ``.so`` files);
2. cythonize the two extensions ``import pysteps`` needs (``motion/_vet.pyx``,
   ``motion/_proesmans.pyx`` -- ``motion/vet.py:43``, ``motion/proesmans.py:18``); the reference's
   own ``setup.py`` cannot run offline (its ``setup_requires`` fetches);
3. write two ~10-line stand-ins for the absent third-party modules ``jsmin`` and ``jsonschema``
   (``pysteps/__init__.py:7-8``): strip ``//`` comments / skip schema validation.  They are only
   ever on ``sys.path`` together with ``oracle/_ref``.

With that ``import pysteps`` works and ``nowcasts.extrapolation.forecast``, ``nowcast_main_loop``
and ``nowcasts.steps`` run unmodified.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may use it (through :func:`activate`); the product never does.
"""

import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
PKG = os.path.join(REF_DIR, "pysteps")
STAMP = os.path.join(REF_DIR, ".built")
SOURCE = os.environ.get("PYSTEPS_REFERENCE", "/root/reference")

_JSMIN = '''"""Test-only stand-in for jsmin (absent here): pystepsrc only needs // comments stripped."""
import re


def jsmin(text, **_):
    out = []
    for line in text.splitlines():
        # a // that is not inside a string literal: pystepsrc keeps comments on their own or after a value
        m = re.match(r'^((?:[^"/]|"(?:[^"\\\\]|\\\\.)*"|/(?!/))*)//.*$', line)
        out.append(m.group(1) if m else line)
    return "\\n".join(out)
'''

_JSONSCHEMA = '''"""Test-only stand-in for jsonschema (absent here): no schema validation."""


class Draft4Validator:
    def __init__(self, schema, *a, **k):
        self.schema = schema

    def iter_errors(self, instance):
        return iter(())

    def validate(self, instance):
        return None
'''


def available():
    """True if oracle/_ref holds an importable pysteps copy."""
    return os.path.exists(STAMP) and os.path.isdir(PKG)


def _ext_suffix():
    return sysconfig.get_config_var("EXT_SUFFIX") or ".so"


def build(force=False):
    """Returns the path of oracle/_ref (or None when neither the reference nor a previous build exists)."""
    if available() and not force:
        return REF_DIR
    if not os.path.isdir(os.path.join(SOURCE, "pysteps")):
        return REF_DIR if available() else None
    import numpy

    if os.path.isdir(REF_DIR):
        shutil.rmtree(REF_DIR)
    os.makedirs(REF_DIR)
    shutil.copytree(
        os.path.join(SOURCE, "pysteps"),
        PKG,
        ignore=shutil.ignore_patterns("__pycache__", "*.pyc"),
    )
    shims = os.path.join(REF_DIR, "_shims")
    os.makedirs(shims)
    with open(os.path.join(shims, "jsmin.py"), "w") as f:
        f.write(_JSMIN)
    with open(os.path.join(shims, "jsonschema.py"), "w") as f:
        f.write(_JSONSCHEMA)
    env = dict(os.environ)
    env["CFLAGS"] = "-fopenmp -O3 -I%s %s" % (numpy.get_include(), env.get("CFLAGS", ""))
    env["LDFLAGS"] = "-fopenmp " + env.get("LDFLAGS", "")
    pyx = [os.path.join("pysteps", "motion", "_vet.pyx"), os.path.join("pysteps", "motion", "_proesmans.pyx")]
    subprocess.check_call(
        [sys.executable, "-m", "cython", "-3"] + pyx, cwd=REF_DIR, env=env
    )
    inc = sysconfig.get_paths()["include"]
    for p in pyx:
        c = os.path.join(REF_DIR, p[:-4] + ".c")
        so = os.path.join(REF_DIR, p[:-4] + _ext_suffix())
        subprocess.check_call(
            ["gcc", "-w", "-shared", "-fPIC", "-O2", "-fopenmp", "-I" + inc, "-I" + numpy.get_include(), c, "-o", so, "-lm"]
        )
        os.remove(c)
    # same configuration file, banner switched off (pysteps/__init__.py:192)
    with open(os.path.join(PKG, "pystepsrc")) as f:
        rc = f.read().replace('"silent_import": false', '"silent_import": true')
    with open(os.path.join(REF_DIR, "pystepsrc"), "w") as f:
        f.write(rc)
    with open(STAMP, "w") as f:
        f.write("built from %s\n" % SOURCE)
    return REF_DIR


def activate():
    """Put oracle/_ref (+ the two stand-ins) on sys.path and import the real pysteps.  Test use only."""
    if not available():
        raise ImportError("oracle/_ref is not built (python -m oracle.build_ref needs /root/reference)")
    stub = sys.modules.get("pysteps")
    if stub is not None and getattr(stub, "__pysteps_reference_stub__", False):
        # tools/ref_loader.py's path stubs: drop them, the real package replaces them
        for k in [k for k in sys.modules if k == "pysteps" or k.startswith("pysteps.")]:
            del sys.modules[k]
    for p in (os.path.join(REF_DIR, "_shims"), REF_DIR):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("PYSTEPSRC", os.path.join(REF_DIR, "pystepsrc"))
    import pysteps

    return pysteps


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
