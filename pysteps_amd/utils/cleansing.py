"""Quality control of sparse motion vectors (host side, vectorised NumPy).

Mirrors pysteps/utils/cleansing.py: ``decluster`` (:21-121) and
``detect_outliers`` (:124-249) with the same signatures, return values and
exceptions.  The reference walks the samples in Python loops (0.1-0.2 s for the
~2000 vectors dense LK pools); here every sample is processed at once.  The data
are a few thousand 2-vectors, far below the size where a device kernel pays, so
this stage is host control logic between the tracker and the IDW kernel.
"""

import warnings

import numpy as np

__all__ = ["decluster", "detect_outliers", "detect_outliers_device"]


def decluster(coord, input_array, scale, min_samples=1, verbose=False):
    """Replace the samples of every ``scale``-sized grid cell by their median.

    Same contract as the reference (:21-75 doc): returns ``(out_coord (l,d),
    output_array (l,m))`` ordered lexicographically by cell index; cells with
    fewer than ``min_samples`` samples are dropped.
    """
    coord = np.array(coord, dtype=float, copy=True)
    input_array = np.array(input_array, dtype=float, copy=True)
    if np.any(~np.isfinite(input_array)):
        raise ValueError("input_array contains non-finite values")
    if input_array.ndim == 1:
        input_array = input_array[:, None]
    elif input_array.ndim != 2:
        raise ValueError(
            "input_array must have 1 (n) or 2 dimensions (n, m), but it has %i" % input_array.ndim
        )
    if coord.ndim != 2:
        raise ValueError("coord must have 2 dimensions (n, d), but it has %i" % coord.ndim)
    if coord.shape[0] != input_array.shape[0]:
        raise ValueError(
            "the number of samples in the input_array does not match the "
            + "number of coordinates %i!=%i" % (input_array.shape[0], coord.shape[0])
        )
    if np.isscalar(scale):
        scale = float(scale)
    else:
        scale = np.array(scale, dtype=float)
        if scale.ndim != 1:
            raise ValueError("scale must have 1 dimension (d), but it has %i" % scale.ndim)
        if scale.shape[0] != coord.shape[1]:
            raise ValueError(
                "scale must have %i elements, but it has %i" % (coord.shape[1], scale.shape[0])
            )
        scale = scale[None, :]

    nvar, ndim = input_array.shape[1], coord.shape[1]
    if coord.shape[0] == 0:
        return np.empty((0, ndim)), np.empty((0, nvar))

    if nvar == 2 and ndim == 2 and isinstance(scale, float) and np.all(np.isfinite(coord)):
        # the shape dense LK uses: native host routine of the C ABI (psh_decluster_host)
        native = _decluster_native(coord, input_array, scale, min_samples)
        if native is not None:
            if verbose:
                print("--- %i samples left after declustering ---" % native[1].shape[0])
            return native

    cells = np.floor(coord / scale)
    group, counts = _group_rows(cells)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    lo = starts + (counts - 1) // 2
    hi = starts + counts // 2

    def cell_medians(columns):
        out = np.empty((counts.size, columns.shape[1]))
        for c in range(columns.shape[1]):
            ordered = columns[np.lexsort((columns[:, c], group)), c]
            out[:, c] = 0.5 * (ordered[lo] + ordered[hi])
        return out

    keep = counts >= min_samples
    dinput = cell_medians(input_array)[keep]
    dcoord = cell_medians(coord)[keep]
    if verbose:
        print("--- %i samples left after declustering ---" % dinput.shape[0])
    return dcoord, dinput


def _decluster_native(coord, values, scale, min_samples):
    """psh_decluster_host (pure host C++, no GPU needed); None if the library is not built."""
    import ctypes  # noqa: PLC0415

    from .. import _lib  # noqa: PLC0415

    try:
        lib = _lib.load()
    except _lib.HipLibraryError:
        return None
    n = coord.shape[0]
    xy = np.ascontiguousarray(coord, dtype=np.float64)
    uv = np.ascontiguousarray(values, dtype=np.float64)
    oxy, ouv = np.empty((n, 2)), np.empty((n, 2))
    count = ctypes.c_int(0)
    rc = lib.psh_decluster_host(xy.ctypes.data, uv.ctypes.data, n, float(scale), int(min_samples),
                                oxy.ctypes.data, ouv.ctypes.data, ctypes.byref(count))
    _lib.check(rc, "psh_decluster_host")
    return oxy[: count.value].copy(), ouv[: count.value].copy()


def _group_rows(cells):
    """Group id per row (ids follow the lexicographic row order of np.unique(axis=0))
    and the size of every group.  Integer-valued rows are folded into one mixed-radix
    int64 key, which sorts an order of magnitude faster than structured rows."""
    lo, hi = cells.min(axis=0), cells.max(axis=0)
    span = hi - lo + 1.0
    if np.all(np.isfinite(span)) and np.prod(span) < 2.0**62:
        key = np.zeros(cells.shape[0], dtype=np.int64)
        for c in range(cells.shape[1]):
            key = key * np.int64(span[c]) + (cells[:, c] - lo[c]).astype(np.int64)
        _, group, counts = np.unique(key, return_inverse=True, return_counts=True)
    else:
        _, group, counts = np.unique(cells, axis=0, return_inverse=True, return_counts=True)
    return group.ravel(), counts


def _knn_indices(coord, k):
    """Indices of the k nearest samples of every sample (itself first).

    With SciPy present the query goes through ``cKDTree`` like the reference
    (:198,221-222), so that ties between equidistant neighbours - common for
    integer feature positions - resolve identically; otherwise brute force.
    """
    try:
        from scipy.spatial import cKDTree  # noqa: PLC0415

        _, inds = cKDTree(coord).query(coord, k=k)
        return inds.reshape(coord.shape[0], k)
    except ImportError:
        pass
    n = coord.shape[0]
    out = np.empty((n, k), dtype=np.intp)
    block = max(1, int(4e6 // max(n, 1)))
    sq = np.einsum("ij,ij->i", coord, coord)
    for s in range(0, n, block):
        e = min(n, s + block)
        d2 = sq[s:e, None] + sq[None, :] - 2.0 * coord[s:e] @ coord.T
        # the sample itself must come first, whatever the rounding of the expansion
        d2[np.arange(e - s), np.arange(s, e)] = -1.0
        if k < n:
            part = np.argpartition(d2, k - 1, axis=1)[:, :k]
        else:
            part = np.broadcast_to(np.arange(n), (e - s, n)).copy()
        rows = np.arange(e - s)[:, None]
        out[s:e] = part[rows, np.argsort(d2[rows, part], axis=1, kind="stable")]
    return out


def detect_outliers(input_array, thr, coord=None, k=None, verbose=False):
    """Flag samples more than ``thr`` standard deviations (Mahalanobis distance for
    multivariate data) from the mean of all samples, or of their ``k`` nearest
    neighbours when ``coord`` and ``k`` are given.  Contract of the reference (:124-249).
    """
    input_array = np.array(input_array, dtype=float, copy=True)
    if np.any(~np.isfinite(input_array)):
        raise ValueError("input_array contains non-finite values")
    if input_array.ndim == 1:
        nsamples, nvar = input_array.size, 1
    elif input_array.ndim == 2:
        nsamples, nvar = input_array.shape
    else:
        raise ValueError(
            f"input_array must have 1 (n) or 2 dimensions (n, m), but it has {input_array.ndim}"
        )
    if nsamples < 2:
        return np.zeros(nsamples, dtype=bool)

    local = coord is not None and k is not None
    if local:
        coord = np.array(coord, dtype=float, copy=True)
        if coord.ndim == 1:
            coord = coord[:, None]
        elif coord.ndim > 2:
            raise ValueError(f"coord must have 2 dimensions (n, d),but it has {coord.ndim}")
        if coord.shape[0] != nsamples:
            raise ValueError(
                "the number of samples in input_array does not match the "
                f"number of coordinates {nsamples}!={coord.shape[0]}"
            )
        k = int(min(nsamples, k + 1))

    with np.errstate(divide="ignore", invalid="ignore"):
        if not local:
            if nvar == 1:
                z = np.abs(input_array - np.mean(input_array)) / np.std(input_array)
                outliers = z > thr
            else:
                z = input_array - np.mean(input_array, axis=0)
                try:
                    vi = np.linalg.inv(np.cov(z.T))
                    md = np.sqrt(np.einsum("ij,jk,ik->i", z, vi, z))
                except np.linalg.LinAlgError as err:
                    warnings.warn(f"{err} during outlier detection")
                    md = np.zeros(nsamples)
                outliers = md > thr
        else:
            nb = _knn_indices(coord, k)[:, 1:]  # neighbours without the sample itself
            if nvar == 1:
                data = input_array.reshape(nsamples)
                neigh = data[nb]
                z = np.abs(data - neigh.mean(axis=1)) / neigh.std(axis=1)
                outliers = z > thr
            else:
                neigh = input_array[nb]  # (n, k-1, nvar)
                mean = neigh.mean(axis=1)
                z = input_array - mean
                dev = neigh - mean[:, None, :]
                dof = max(neigh.shape[1] - 1, 0)
                cov = np.einsum("nki,nkj->nij", dev, dev) / dof if dof > 0 else np.full(
                    (nsamples, nvar, nvar), np.nan
                )
                md = np.zeros(nsamples)
                ok = np.isfinite(cov).all(axis=(1, 2))
                if nvar == 2:
                    ok &= (cov[:, 0, 0] * cov[:, 1, 1] - cov[:, 0, 1] * cov[:, 1, 0]) != 0.0
                else:
                    ok &= np.linalg.matrix_rank(np.where(ok[:, None, None], cov, 0.0)) == nvar
                if not ok.all():
                    warnings.warn("Singular matrix during outlier detection")
                if ok.any():
                    vi = np.linalg.inv(cov[ok])
                    md[ok] = np.sqrt(np.einsum("ni,nij,nj->n", z[ok], vi, z[ok]))
                outliers = md > thr
    outliers = np.asarray(outliers, dtype=bool).reshape(nsamples)
    if verbose:
        print(f"--- {np.sum(outliers)} outliers detected ---")
    return outliers


def detect_outliers_device(input_array, thr, coord, k, verbose=False):
    """Local multivariate test of :func:`detect_outliers` for 2-vectors with 2-d
    coordinates, evaluated by the HIP kernel ``csrc/sparse_qc.hip`` (float64, brute
    force k-NN, ties -> lower index).  Same result as the host version except where
    the k-th and (k+1)-th neighbours are exactly equidistant (cKDTree's tie order is
    unspecified).  Other input shapes are routed to the host implementation."""
    from .. import _lib  # noqa: PLC0415

    values = np.ascontiguousarray(input_array, dtype=np.float64)
    xy = None if coord is None else np.ascontiguousarray(coord, dtype=np.float64)
    if (
        values.ndim != 2 or values.shape[1] != 2 or xy is None or k is None
        or xy.shape != values.shape or values.shape[0] > 8192
    ):
        return detect_outliers(input_array, thr, coord, k, verbose)
    if np.any(~np.isfinite(values)):
        raise ValueError("input_array contains non-finite values")
    n = values.shape[0]
    flags = np.zeros(n, dtype=np.uint8)
    if n >= 2:
        rc = _lib.lib().psh_outliers_local_host(xy.ctypes.data, values.ctypes.data, n, int(k), float(thr),
                                                flags.ctypes.data)
        _lib.check(rc, "psh_outliers_local_host")
    out = flags.astype(bool)
    if verbose:
        print(f"--- {np.sum(out)} outliers detected ---")
    return out
