"""The oracle restatement driven by the REAL pysteps callers (CPU; oracle pinning).

``oracle/semilag.py`` is registered in the reference's own extrapolation table and the callers of
SURVEY 8a row a14 run it by name next to the stock operator: the restatement honours every calling
convention the callers use (``xy_coords=``, positional ``outval="min"``, ``displacement_prev``
threading, the displacement-only call with ``precip=None``) and agrees with the reference to
float64 round-off THROUGH those callers, not only on direct calls.
"""

import numpy as np
import pytest

from conftest import nan_mismatch


@pytest.fixture(scope="module")
def pysteps(ref_pysteps):
    from oracle import semilag as osl
    from pysteps import extrapolation

    extrapolation.interface._extrapolation_methods["semilagrangian_oracle"] = osl.extrapolate
    yield ref_pysteps
    extrapolation.interface._extrapolation_methods.pop("semilagrangian_oracle", None)


def test_reference_package_is_the_unmodified_reference(ref_pysteps):
    """oracle/_ref imports, its compiled Cython modules load, the reference's own KATs pass."""
    import pysteps
    from pysteps import extrapolation, motion, nowcasts

    assert "oracle/_ref" in pysteps.__file__.replace("\\", "/")
    assert callable(motion.get_method("vet")) and callable(motion.get_method("proesmans"))
    assert callable(nowcasts.get_method("steps"))
    ex = extrapolation.get_method("semilagrangian")
    # pysteps/tests/test_extrapolation_semilagrangian.py:9-24
    precip = np.zeros((8, 8))
    precip[0, 0] = 1
    result = ex(precip, np.ones((2, 8, 8)), 1)[0]
    expected = np.zeros((8, 8))
    expected[:, 0] = np.nan
    expected[0, :] = np.nan
    expected[1, 1] = 1
    np.testing.assert_array_equal(result, expected)


def test_extrapolation_nowcast_through_the_real_caller(pysteps):
    from pysteps import nowcasts
    from tools import synth

    m, n = 160, 192
    P = synth.rain_field_db(m, n, seed=11, sigma=2.0)
    P[synth.border_nan_mask(m, n, 0.1)] = np.nan
    V = synth.true_velocity(m, n)
    fc = nowcasts.get_method("extrapolation")
    want = fc(P, V, 6, extrap_method="semilagrangian")
    got = fc(P, V, 6, extrap_method="semilagrangian_oracle")
    assert nan_mismatch(got, want) == 0
    assert np.nanmax(np.abs(got - want)) < 1e-5


@pytest.mark.parametrize("method,kw", [
    ("steps", dict(n_ens_members=2, n_cascade_levels=4, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, seed=42)),
    ("sprog", dict(n_cascade_levels=4, precip_thr=-10.0)),
])
def test_main_loop_callers_with_the_oracle(pysteps, method, kw):
    from pysteps import nowcasts
    from tools import synth

    frames = synth.steps_frames(128, 128, 3)
    V = synth.true_velocity(128, 128).astype(np.float64)
    fn = nowcasts.get_method(method)
    want = fn(frames, V, [0.5, 1.0, 2.0], extrap_method="semilagrangian", **kw)
    got = fn(frames, V, [0.5, 1.0, 2.0], extrap_method="semilagrangian_oracle", **kw)
    assert got.shape == want.shape
    assert nan_mismatch(got, want) == 0
    assert np.nanmax(np.abs(got - want)) < 1e-3
    assert np.count_nonzero(np.abs(got - want) > 1e-6) < 1e-3 * want.size
