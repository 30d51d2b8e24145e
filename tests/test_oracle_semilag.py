"""Pin the CPU oracles (oracle/semilag.py, oracle/semilag_c.c) to the reference.

* the two known-answer tests of pysteps/tests/test_extrapolation_semilagrangian.py:9-24,57-72
* golden vectors produced by the unmodified reference (tools/make_golden.py)
* if /root/reference is present (build container), a live comparison as well
"""

import numpy as np
import pytest
from numpy.testing import assert_array_almost_equal

from oracle import semilag as osl
from oracle import semilag_cport as ocl
from tools import ref_loader

from conftest import nan_mismatch

GOLDEN_SL = [
    "sl_int_T6", "sl_shear_K3", "sl_K0", "sl_list_vt", "sl_nan_min", "sl_nan_nan",
    "sl_order0", "sl_resume", "sl_resume_K0", "sl_f64", "sl_order3", "sl_order3_nan",
    "sl_mode_nearest", "sl_mode_reflect_nan", "sl_mode_mirror", "sl_mode_wrap", "sl_mode_gridwrap",
    "sl_mode_gridconst", "sl_mode_reflect_o0", "sl_mode_gridwrap_o0",
    "sl_velnan", "sl_velnan_K3_nanfield", "sl_velnan_K0_o0", "sl_velnan_min",
]
BOUNDARY_MODES = ["constant", "nearest", "reflect", "mirror", "wrap", "grid-constant", "grid-wrap"]


def _kat_inputs(speed):
    precip = np.zeros((8, 8))
    precip[0, 0] = 1
    v = np.ones((8, 8)) * speed
    expected = np.zeros((1, 8, 8))
    expected[:, :, 0] = np.nan
    expected[:, 0, :] = np.nan
    expected[:, 1, 1] = 1
    return precip, np.stack([v, v]), expected


@pytest.mark.parametrize("backend", ["numpy", "scipy", "c"])
@pytest.mark.parametrize("speed,timesteps", [(1, 1), (10, [0.1])])
def test_reference_known_answers(backend, speed, timesteps):
    precip, velocity, expected = _kat_inputs(speed)
    if backend == "c":
        result = ocl.extrapolate(precip, velocity, timesteps)
    else:
        result = osl.extrapolate(precip, velocity, timesteps, backend=backend)
    assert_array_almost_equal(result, expected)


def _run(backend, c):
    kw = dict(c["kw"])
    allow = bool(kw.pop("allow_nonfinite_values", False)) or not np.all(np.isfinite(c["precip"]))
    if backend == "c":
        return ocl.extrapolate(c["precip"], c["velocity"], c["timesteps"], return_displacement=True, **kw)
    return osl.extrapolate(c["precip"], c["velocity"], c["timesteps"], return_displacement=True,
                           allow_nonfinite_values=allow, backend=backend, **kw)


@pytest.mark.parametrize("backend", ["numpy", "scipy", "c"])
@pytest.mark.parametrize("name", GOLDEN_SL)
def test_oracle_matches_reference_golden(semilag_golden, backend, name):
    c = semilag_golden.case(name)
    if backend == "c" and c["precip"].dtype != np.float32:
        pytest.skip("the C port takes float32 fields")
    if backend == "c" and c["kw"].get("interp_order", 1) > 1:
        pytest.skip("the C port restates interpolation order 0/1")
    if backend == "c" and "map_coordinates_mode" in c["kw"]:
        pytest.skip("the C port restates mode constant")
    out, disp = _run(backend, c)
    assert out.shape == c["out"].shape
    if backend != "c":
        assert out.dtype == c["out"].dtype
    assert nan_mismatch(out, c["out"]) == 0
    np.testing.assert_allclose(disp, c["disp"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out, c["out"], rtol=0, atol=2e-6, equal_nan=True)


@pytest.mark.parametrize("backend", ["numpy", "c"])
def test_displacement_only(semilag_golden, backend):
    c = semilag_golden.case("sl_disp_only")
    mod = ocl if backend == "c" else osl
    none, disp = mod.extrapolate(None, c["velocity"], [0.7], return_displacement=True, n_iter=1)
    assert none is None
    np.testing.assert_allclose(disp, c["disp"], rtol=0, atol=1e-11)


GOLDEN_XY = ["sl_xy_warp", "sl_xy_warp_resume", "sl_xy_half_K0", "sl_xy_warp_o0", "sl_xy_warp_o3"]


@pytest.mark.parametrize("backend", ["numpy", "scipy"])
@pytest.mark.parametrize("name", GOLDEN_XY)
def test_oracle_custom_grid_matches_reference_golden(semilag_xy_golden, backend, name):
    """custom xy_coords (reference :174-179): the oracle against outputs of the unmodified reference"""
    c = semilag_xy_golden.case(name)
    out, disp = _run(backend, c)
    assert out.shape == c["out"].shape and out.dtype == c["out"].dtype
    assert nan_mismatch(out, c["out"]) == 0
    np.testing.assert_allclose(disp, c["disp"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out, c["out"], rtol=0, atol=2e-6, equal_nan=True)


@pytest.mark.parametrize("backend,atol", [("scipy", 1e-11), ("numpy", 2e-7)])
def test_oracle_custom_grid_displacement_only(semilag_xy_golden, backend, atol):
    """(float32 velocity: SciPy rounds every interpolated velocity sample to float32, the NumPy restatement keeps
    float64 - 8e-8 px over two lead steps)"""
    c = semilag_xy_golden.case("sl_xy_disp_only")
    none, disp = osl.extrapolate(None, c["velocity"], [0.7, 1.9], return_displacement=True, n_iter=1, xy_coords=c["xy_coords"],
                                 backend=backend)
    assert none is None
    np.testing.assert_allclose(disp, c["disp"], rtol=0, atol=atol)


GOLDEN_ORDERS = ["sl_o2", "sl_o2_nan", "sl_o2_reflect", "sl_o4", "sl_o4_nan", "sl_o4_nearest", "sl_o4_gridconstant_nan",
                 "sl_o5", "sl_o5_nan", "sl_o5_gridwrap"]


@pytest.mark.parametrize("backend", ["numpy", "scipy"])
@pytest.mark.parametrize("name", GOLDEN_ORDERS)
def test_oracle_spline_orders_match_reference_golden(semilag_orders_golden, backend, name):
    """interp_order 2 / 4 / 5: the restated prefilter poles, tap geometry (even orders start at floor(c + 0.5) -
    order // 2) and B-spline weights against outputs of the unmodified reference"""
    c = semilag_orders_golden.case(name)
    out, disp = _run(backend, c)
    assert out.shape == c["out"].shape and out.dtype == c["out"].dtype
    assert nan_mismatch(out, c["out"]) == 0
    np.testing.assert_allclose(disp, c["disp"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out, c["out"], rtol=0, atol=5e-5, equal_nan=True)


@pytest.mark.parametrize("order", [2, 4, 5])
def test_spline_restatement_equals_scipy(order):
    """prefilter and sampler of the NumPy backend against scipy.ndimage itself, every boundary mode"""
    from scipy.ndimage import map_coordinates, spline_filter

    rng = np.random.default_rng(order)
    f = rng.standard_normal((37, 53))
    for kind, mode in (("mirror", "mirror"), ("reflect", "reflect"), ("wrap", "grid-wrap")):
        np.testing.assert_allclose(osl._spline_prefilter(f, kind, order), spline_filter(f, order=order, mode=mode), rtol=0, atol=1e-10)
    row, col = rng.uniform(-3, 40, 3000), rng.uniform(-3, 56, 3000)
    for mode in ("constant", "nearest", "reflect", "mirror", "wrap", "grid-wrap", "grid-constant"):
        np.testing.assert_allclose(osl._numpy_sample(f, row, col, mode, -7.0, order),
                                   map_coordinates(f, [row, col], order=order, mode=mode, cval=-7.0), rtol=0, atol=1e-11)


def test_oracle_error_behaviour():
    p = np.ones((8, 8))
    v = np.ones((2, 8, 8))
    with pytest.raises(ValueError):
        osl.extrapolate(np.ones(8), v, 1)
    with pytest.raises(ValueError):
        osl.extrapolate(p, np.ones((8, 8)), 1)
    with pytest.raises(ValueError):
        osl.extrapolate(p, v, [1, 2, 3, 5, 4, 6, 7])
    with pytest.raises(ValueError):
        osl.extrapolate(None, v, 1)


def test_chained_calls_equal_one_call():
    """displacement_prev chaining is bitwise equal to a multi-step call."""
    rng = np.random.default_rng(5)
    p = rng.random((40, 56)).astype(np.float32)
    y, x = np.mgrid[0:40, 0:56]
    v = np.stack([2 + 0.05 * y, -1 + 0.04 * x]).astype(np.float32)
    full, dfull = osl.extrapolate(p, v, 3, return_displacement=True)
    d = None
    for t in range(3):
        out, d = osl.extrapolate(p, v, [1.0], return_displacement=True, displacement_prev=d)
        assert np.array_equal(out[0], full[t], equal_nan=True)
    assert np.array_equal(d, dfull)


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
def test_oracle_matches_live_reference():
    ref = ref_loader.load("pysteps.extrapolation.semilagrangian")
    rng = np.random.default_rng(11)
    m, n = 50, 70
    p = rng.gamma(1.0, 2.0, (m, n)).astype(np.float32)
    y, x = np.mgrid[0:m, 0:n]
    v = np.stack([3 + 0.08 * (y - m / 2), -2 + 0.06 * (x - n / 2)]).astype(np.float32)
    for k in (0, 1, 2):
        r, rd = ref.extrapolate(p, v, [0.5, 1.25, 2.0], n_iter=k, vel_timestep=0.5, return_displacement=True)
        for mod, kw in ((osl, dict(backend="numpy")), (ocl, {})):
            a, ad = mod.extrapolate(p, v, [0.5, 1.25, 2.0], n_iter=k, vel_timestep=0.5,
                                    return_displacement=True, **kw)
            assert nan_mismatch(a, r) == 0
            np.testing.assert_allclose(ad, rd, rtol=0, atol=1e-11)
            np.testing.assert_allclose(a, r, rtol=0, atol=2e-6, equal_nan=True)


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("mode", BOUNDARY_MODES)
def test_boundary_modes_pinned_against_scipy(mode, order):
    """The restated folding rules of oracle.semilag (coordinate, then tap by tap) against
    scipy.ndimage.map_coordinates itself: lattice and random coordinates several periods outside
    the array, NaN planted at every index in turn (a NaN tap poisons the sample even at weight 0,
    so the NaN pattern reveals WHICH index an out-of-range tap was folded to)."""
    from scipy.ndimage import map_coordinates

    rng = np.random.default_rng(17)
    for m, n in ((1, 1), (1, 4), (2, 2), (3, 5), (5, 3)):
        lat_r = np.arange(-3 * m - 2, 3 * m + 2.01, 0.25)
        lat_c = np.arange(-3 * n - 2, 3 * n + 2.01, 0.25)
        rows = np.concatenate([np.repeat(lat_r, lat_c.size), rng.uniform(-3 * m - 2, 3 * m + 2, 300)])
        cols = np.concatenate([np.tile(lat_c, lat_r.size), rng.uniform(-3 * n - 2, 3 * n + 2, 300)])
        for nan_at in [None] + list(range(m * n)):
            field = rng.uniform(1.0, 9.0, (m, n))
            if nan_at is not None:
                field.flat[nan_at] = np.nan
            want = map_coordinates(field, [rows, cols], order=order, mode=mode, cval=-7.0, prefilter=False)
            got = osl._numpy_sample(field, rows, cols, mode, -7.0, order)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (mode, order, m, n, nan_at)
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, equal_nan=True)


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
@pytest.mark.parametrize("mode", BOUNDARY_MODES[1:])
def test_oracle_boundary_modes_match_live_reference(mode):
    ref = ref_loader.load("pysteps.extrapolation.semilagrangian")
    rng = np.random.default_rng(23)
    m, n = 40, 56
    p = rng.gamma(1.0, 2.0, (m, n)).astype(np.float32)
    p[rng.uniform(size=(m, n)) < 0.03] = np.nan
    y, x = np.mgrid[0:m, 0:n]
    v = np.stack([3 + 0.08 * (y - m / 2), -2 + 0.06 * (x - n / 2)]).astype(np.float32)
    for order in (0, 1):
        r = ref.extrapolate(p, v, [2.0, 9.0, 31.0], interp_order=order, map_coordinates_mode=mode, outval=-3.0,
                            allow_nonfinite_values=True)
        a = osl.extrapolate(p, v, [2.0, 9.0, 31.0], interp_order=order, map_coordinates_mode=mode, outval=-3.0,
                            allow_nonfinite_values=True)
        assert nan_mismatch(a, r) == 0
        np.testing.assert_allclose(a, r, rtol=0, atol=2e-6, equal_nan=True)


O3_MODES = ("nearest", "reflect", "mirror", "wrap", "grid-wrap", "grid-constant")
GOLDEN_O3 = (["sl_o3_%s%s" % (m.replace("-", ""), suffix) for m in O3_MODES for suffix in ("", "_nan")]
             + ["sl_o3_gridconstant_nancval"])


@pytest.mark.parametrize("mode", ("constant",) + O3_MODES)
def test_spline_prefilter_kinds_pinned_against_scipy(mode):
    """The boundary initialisation of SciPy's spline filter per map_coordinates mode (mirror / reflect /
    periodic) as restated in the oracle, against scipy.ndimage.spline_filter itself."""
    from scipy.ndimage import spline_filter

    rng = np.random.default_rng(5)
    for shape in ((16, 23), (40, 33), (9, 70)):
        a = rng.standard_normal(shape)
        got = osl._spline_prefilter(a, osl._PREFILTER_KIND[mode])
        want = spline_filter(a, order=3, mode=mode, output=np.float64)
        # (the reflect initialisation is exact from ~16 samples on: its z^n terms)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 if min(shape) < 16 else 1e-13)


@pytest.mark.parametrize("mode", O3_MODES)
def test_order3_boundary_modes_pinned_against_scipy(mode):
    """order-3 map_coordinates with a boundary mode: padding for the two modes without a boundary condition in
    the filter, coordinate folding on the original lengths, taps folded on the padded ones - against SciPy
    on coordinates up to several array lengths outside."""
    from scipy.ndimage import map_coordinates

    rng = np.random.default_rng(11)
    for (m, n) in ((17, 21), (40, 33)):
        f = rng.standard_normal((m, n))
        row = rng.uniform(-2.5 * m, 3.5 * m, 3000)
        col = rng.uniform(-2.5 * n, 3.5 * n, 3000)
        row[:1500] = rng.uniform(-2, m + 1, 1500)
        col[:1500] = rng.uniform(-2, n + 1, 1500)
        row[:50] = np.round(row[:50])  # integer coordinates, the edges among them
        col[25:75] = np.round(col[25:75])
        want = map_coordinates(f, [row, col], order=3, mode=mode, cval=-7.5, prefilter=True)
        got = osl._numpy_sample(f, row, col, mode, -7.5, 3)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 if m < 20 and mode == "reflect" else 1e-12)


@pytest.mark.parametrize("name", GOLDEN_O3)
def test_oracle_order3_modes_match_reference_golden(semilag_o3_golden, name):
    c = semilag_o3_golden.case(name)
    out, disp = osl.extrapolate(c["precip"], c["velocity"], c["timesteps"], return_displacement=True, backend="numpy", **c["kw"])
    assert out.shape == c["out"].shape and out.dtype == c["out"].dtype
    assert nan_mismatch(out, c["out"]) == 0
    np.testing.assert_allclose(disp, c["disp"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out, c["out"], rtol=0, atol=2e-6, equal_nan=True)
