import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")
    # a fast path that fails while it builds its device state is an ERROR under test, not a silent fallback to the
    # reference (pysteps_amd/nowcasts/steps_resident.py::try_create); the one test of the fallback clears it
    os.environ.setdefault("PYSTEPS_HIP_STRICT", "1")


def _gpu_present():
    # the kernel driver's device node: present on every box with an AMD GPU, whatever state the
    # library is in - a GPU box with a broken build must FAIL its gpu tests, not skip them
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """On a host without a gfx950 device the gpu-marked tests are skipped, not failed (plain `pytest`
    stays meaningful on CPU-only CI); on a GPU box nothing is skipped - a missing library or device
    there must fail loudly, so the check is only made when a gpu test was collected at all."""
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _gpu_present():
        return
    skip = pytest.mark.skip(reason="no MI355X (gfx950) device visible: gpu-marked test")
    for it in gpu_items:
        it.add_marker(skip)


class GoldenCases:
    """Cases stored by tools/make_golden.py: inputs, kwargs and reference outputs."""

    def __init__(self, path):
        self._z = np.load(path, allow_pickle=False)
        self.names = sorted({k.split("/")[0] for k in self._z.files})

    def case(self, name):
        z = self._z
        c = {"kw": {}}
        for key in z.files:
            if not key.startswith(name + "/"):
                continue
            rest = key[len(name) + 1:]
            if rest.startswith("kw/"):
                v = z[key]
                c["kw"][rest[3:]] = v.item() if v.ndim == 0 else v
            else:
                c[rest] = z[key]
        if "timesteps" in c:
            if bool(c.pop("timesteps_is_int")):
                c["timesteps"] = int(c["timesteps"])
            else:
                c["timesteps"] = [float(t) for t in np.atleast_1d(c["timesteps"])]
        return c


@pytest.fixture(scope="session")
def semilag_golden():
    return GoldenCases(os.path.join(GOLDEN, "semilag_reference.npz"))


@pytest.fixture(scope="session")
def semilag_o3_golden():
    """interp_order=3 x the six other boundary modes, from the unmodified reference (tools/make_golden.py)."""
    return GoldenCases(os.path.join(GOLDEN, "semilag_order3_modes.npz"))


@pytest.fixture(scope="session")
def semilag_orders_golden():
    """interp_order 2 / 4 / 5, from the unmodified reference (tools/make_golden.py semilag_spline_orders)."""
    return GoldenCases(os.path.join(GOLDEN, "semilag_spline_orders.npz"))


@pytest.fixture(scope="session")
def semilag_xy_golden():
    """custom xy_coords grids, from the unmodified reference (tools/make_golden.py semilag_xy)."""
    return GoldenCases(os.path.join(GOLDEN, "semilag_xy_coords.npz"))


@pytest.fixture(scope="session")
def ref_pysteps():
    """The REAL reference package, imported from oracle/_ref (built by ``python -m oracle.build_ref``
    from /root/reference; ships to the GPU box with the snapshot).  Test infrastructure only."""
    from oracle import build_ref

    if not build_ref.available() and build_ref.build() is None:
        pytest.skip("oracle/_ref is not built and /root/reference is absent")
    return build_ref.activate()


def rel_l2(a, b):
    """relative L2 error over jointly finite entries (BASELINE.md section 3 'Parity')."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ok = np.isfinite(a) & np.isfinite(b)
    den = np.linalg.norm(b[ok])
    return float(np.linalg.norm(a[ok] - b[ok]) / (den if den > 0 else 1.0))


def nan_mismatch(a, b):
    return int(np.count_nonzero(np.isnan(np.asarray(a)) != np.isnan(np.asarray(b))))
