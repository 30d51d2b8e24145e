"""rbfinterp2d on the device (csrc/rbf.hip) against the reference's own function.

Reference: pysteps/utils/interpolate.py:117-170 (a wrapper of scipy.interpolate.Rbf) behind the preamble
of pysteps/decorators.py:153-250, imported from oracle/_ref.  The weights here are SciPy's own solve (the same
call the reference makes), the device evaluates  sum_j w_j phi(|x - x_j|)  in float64: the two results
differ by the order of a float64 sum whose terms alternate in sign - bar: relative L2 <= 1e-9, observed
values in gpurun_out/rbf_seen.json.  Mirrors pysteps/tests/test_utils_interpolate.py: shapes, finiteness,
single sample / uniform values -> uniform field, error cases."""

import json
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

SEEN = []


@pytest.fixture(scope="module")
def rbf():
    from pysteps_amd.utils import rbfinterp2d

    yield rbfinterp2d
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/rbf_seen.json", "w") as fh:
            json.dump(SEEN, fh)
    except OSError:
        pass


def _samples(n_samples, shape, seed, nvar=2):
    rng = np.random.default_rng(seed)
    m, n = shape
    # declustered-like positions: one sample per cell of a coarse lattice, jittered
    side = int(np.ceil(np.sqrt(n_samples)))
    cells = rng.permutation(side * side)[:n_samples]
    cy, cx = np.divmod(cells, side)
    xy = np.column_stack([(cx + rng.random(n_samples)) * (n - 1) / side, (cy + rng.random(n_samples)) * (m - 1) / side])
    values = rng.normal(0.0, 2.0, (n_samples, nvar)) + np.array([2.0, -1.0, 0.5][:nvar])
    return xy, values if nvar > 1 else values[:, 0]


@pytest.mark.parametrize("function", ["multiquadric", "inverse", "gaussian", "linear", "cubic", "quintic", "thin_plate"])
def test_every_basis_function_matches_the_reference(rbf, ref_pysteps, function):
    from pysteps.utils.interpolate import rbfinterp2d as ref

    shape = (97, 131)
    xy, values = _samples(60, shape, seed=len(function))
    xgrid, ygrid = np.arange(shape[1], dtype=float), np.arange(shape[0], dtype=float)
    want = ref(xy, values, xgrid, ygrid, function=function)
    got = rbf(xy, values, xgrid, ygrid, function=function)
    assert got.shape == want.shape == (2,) + shape and got.dtype == np.float64 and np.isfinite(got).all()
    err = rel_l2(got, want)
    SEEN.append((function, err))
    assert err <= 1e-9, err


@pytest.mark.parametrize("nvar,n_samples,kwargs", [(2, 40, {}), (2, 300, {"epsilon": 7.5}), (3, 80, {"smooth": 0.1}),
                                                   (2, 50, {"function": "gaussian", "epsilon": 30.0, "nchunks": 9})])
def test_variables_options_and_grids(rbf, ref_pysteps, nvar, n_samples, kwargs):
    from pysteps.utils.interpolate import rbfinterp2d as ref

    shape = (64, 200)
    xy, values = _samples(n_samples, shape, seed=n_samples, nvar=nvar)
    xgrid, ygrid = np.linspace(-3.0, 250.0, shape[1]), np.linspace(10.0, 80.0, shape[0])  # any regular grid
    want = ref(xy, values, xgrid, ygrid, **kwargs)
    got = rbf(xy, values, xgrid, ygrid, **kwargs)
    assert got.shape == want.shape
    err = rel_l2(got, want)
    SEEN.append(("nvar%d n%d %s" % (nvar, n_samples, sorted(kwargs)), err))
    assert err <= 1e-9, err


def test_trivial_cases_errors_and_delegation(rbf, ref_pysteps):
    from pysteps.utils.interpolate import rbfinterp2d as ref

    xgrid, ygrid = np.arange(20.0), np.arange(12.0)
    one = rbf(np.array([[3.0, 4.0]]), np.array([[1.5, -2.0]]), xgrid, ygrid)  # decorators.py:200-203
    assert one.shape == (2, 12, 20) and np.all(one[0] == 1.5) and np.all(one[1] == -2.0)
    same = rbf(np.array([[3.0, 4.0], [7.0, 1.0]]), np.array([2.0, 2.0]), xgrid, ygrid)  # :207-208
    assert same.shape == (1, 12, 20) and np.all(same == 2.0)  # (no squeeze in that branch of the reference either)
    xy, values = _samples(20, (12, 20), seed=3)
    bad = values.copy()
    bad[2, 0] = np.nan
    with pytest.raises(ValueError):
        rbf(xy, bad, xgrid, ygrid)
    with pytest.raises(ValueError):
        rbf(xy, values[:-1], xgrid, ygrid)
    with pytest.warns(DeprecationWarning):
        assert rel_l2(rbf(xy, values, xgrid, ygrid, rbfunction="gaussian"), ref(xy, values, xgrid, ygrid, rbfunction="gaussian")) <= 1e-9
    # 1-d values: the reference's own axis handling (a transposition per grid chunk, interpolate.py:169) applies
    with pytest.warns(UserWarning, match="delegating"):
        got = rbf(xy, values[:, 0], np.arange(12.0), ygrid)
    assert np.array_equal(got, ref(xy, values[:, 0], np.arange(12.0), ygrid))
    # a callable basis function and an irregular grid are the reference's job (delegated with a warning)
    irregular = np.array(sorted(np.random.default_rng(0).random(20) * 19.0))
    with pytest.warns(UserWarning, match="delegating"):
        got = rbf(xy, values, irregular, ygrid)
    assert np.allclose(got, ref(xy, values, irregular, ygrid))
    with pytest.warns(UserWarning, match="delegating"):
        got = rbf(xy, values, xgrid, ygrid, function=lambda self, r: r ** 2 + 1.0)
    assert got.shape == (2, 12, 20)


def test_full_size_grid_runs_and_interpolates(rbf):
    """4096 x 4096 with 900 vectors (what dense_lucaskanade hands over after declustering): the interpolant
    goes through its samples."""
    m = n = 4096
    xy, values = _samples(900, (m, n), seed=11)
    xy = np.round(xy)  # on grid nodes, so that the samples can be read back
    _, keep = np.unique(xy, axis=0, return_index=True)
    xy, values = xy[keep], values[keep]
    got = rbf(xy, values, np.arange(n), np.arange(m))
    assert got.shape == (2, m, n) and np.isfinite(got).all()
    back = got[:, xy[:, 1].astype(int), xy[:, 0].astype(int)].T
    assert np.abs(back - values).max() <= 1e-6 * np.abs(values).max()
