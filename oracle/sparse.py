"""CPU oracle for the sparse-vector tail of dense LK.  TEST INFRASTRUCTURE ONLY.

float64 restatements of
  pysteps/utils/cleansing.py:124-249   detect_outliers
  pysteps/utils/cleansing.py:21-121    decluster
  pysteps/utils/interpolate.py:26-114  idwinterp2d  (+ decorators.py:200-208 trivial cases)
following the behavioural spec of SURVEY.md section 8a.  The k-NN queries use
scipy.spatial.cKDTree exactly like the reference (third-party, installed).
Pinned against the unmodified reference functions by tests/test_oracle_sparse.py
(golden fixtures tests/golden/sparse_reference.npz + live comparison when
/root/reference is present).
"""

import numpy as np
from scipy.spatial import cKDTree


def detect_outliers(values, thr, coord, k):
    """Local multivariate / univariate test (the form dense LK uses); ``k=None`` or ``coord=None``:
    the global test against the mean and covariance of all samples (cleansing.py:202-217)."""
    values = np.asarray(values, dtype=float)
    n = values.shape[0]
    if n < 2:
        return np.zeros(n, dtype=bool)
    if k is None or coord is None:
        if values.ndim == 1:
            return np.abs(values - values.mean()) / values.std() > thr
        z = values - values.mean(axis=0)
        try:
            vi = np.linalg.inv(np.cov(z.T))
            md = np.sqrt(np.einsum("ij,jk,ik->i", z, vi, z))
        except np.linalg.LinAlgError:
            md = np.zeros(n)
        return md > thr
    coord = np.asarray(coord, dtype=float)
    kk = min(n, k + 1)
    _, inds = cKDTree(coord).query(coord, k=kk)
    flags = np.zeros(n, dtype=bool)
    for i in range(n):
        nb = values[inds[i, 1:]]
        if values.ndim == 1:
            flags[i] = abs(values[i] - nb.mean()) / nb.std() > thr
            continue
        z = values[i] - nb.mean(axis=0)
        cov = np.cov((nb - nb.mean(axis=0)).T)
        try:
            md = np.sqrt(z @ np.linalg.inv(cov) @ z)
        except np.linalg.LinAlgError:
            md = 0.0
        flags[i] = md > thr
    return flags


def decluster(coord, values, scale, min_samples=1):
    coord = np.asarray(coord, dtype=float)
    values = np.asarray(values, dtype=float)
    if values.ndim == 1:
        values = values[:, None]
    cells = np.floor(coord / scale)
    out_c, out_v = [], []
    for cell in np.unique(cells, axis=0):  # lexicographic order
        member = np.all(cells == cell, axis=1)
        if member.sum() >= min_samples:
            out_v.append(np.median(values[member], axis=0))
            out_c.append(np.median(coord[member], axis=0))
    if not out_c:
        return np.empty((0, coord.shape[1])), np.empty((0, values.shape[1]))
    return np.array(out_c), np.array(out_v)


def idw(xy, values, m, n, k=20, power=0.5, dist_offset=0.5):
    """values (L,2) -> (2,m,n) float64 on the unit grid x=0..n-1, y=0..m-1."""
    xy = np.asarray(xy, dtype=float)
    values = np.asarray(values, dtype=float)
    L = xy.shape[0]
    if L == 1:
        return np.ones((2, m, n)) * values[0][:, None, None]
    if values.max() == values.min():
        return np.ones((2, m, n)) * values.ravel()[0]
    gx, gy = np.meshgrid(np.arange(n), np.arange(m))
    grid = np.column_stack([gx.ravel(), gy.ravel()])
    kk = L if k is None else min(k, L)
    dist, inds = cKDTree(xy).query(grid, k=kk)
    if dist.ndim == 1:
        dist, inds = dist[:, None], inds[:, None]
    w = 1.0 / np.power(dist + dist_offset, power)
    w /= w.sum(axis=1, keepdims=True)
    out = np.sum(values[inds, :] * w[..., None], axis=1)
    return np.moveaxis(out.reshape(m, n, 2), -1, 0)
