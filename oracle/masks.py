"""TEST INFRASTRUCTURE - CPU restatement of pysteps/nowcasts/utils.py:69-101 (``compute_dilated_mask``)
in the form csrc/mask.hip evaluates it: one dilation by the structuring element, then a truncated L1
distance transform instead of ``r`` further dilations by the 4-neighbour cross.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned against the reference itself (scipy.ndimage.binary_dilation underneath): tests/test_masks_cpu.py
runs both on random masks with symmetric, asymmetric and even-sized structuring elements, r = 0 ... 10,
empty masks included - bit-identical.
"""

import numpy as np


def _shift_rows(a, d, fill):
    """out[i] = a[i + d] where that row exists, else fill."""
    out = np.full_like(a, fill)
    m = a.shape[0]
    lo, hi = max(0, -d), min(m, m - d)
    if lo < hi:
        out[lo:hi] = a[lo + d:hi + d]
    return out


def binary_dilation(mask, structure):
    """scipy.ndimage.binary_dilation(mask, structure) with origin 0 and border_value 0:
    out[p] = OR over the set elements s of mask[p - (s - centre)], centre = shape // 2."""
    mask = np.asarray(mask) != 0
    structure = np.asarray(structure) != 0
    cy, cx = structure.shape[0] // 2, structure.shape[1] // 2
    out = np.zeros_like(mask)
    for sy, sx in zip(*np.nonzero(structure)):
        out |= _shift_rows(_shift_rows(mask, -(sy - cy), False).T, -(sx - cx), False).T
    return out


def compute_dilated_mask(input_mask, kr, r):
    mask0 = binary_dilation(np.ndarray.astype(np.asarray(input_mask).copy(), "uint8"), kr)  # :87-88
    cap = r + 1
    g = np.full(mask0.shape, cap, np.int64)  # distance to the nearest set pixel of the column
    for d in range(-r, r + 1):
        g = np.where(_shift_rows(mask0, d, False), np.minimum(g, abs(d)), g)
    dist = np.full(mask0.shape, cap, np.int64)  # L1 distance to mask0, truncated at r + 1
    for d in range(-r, r + 1):
        dist = np.minimum(dist, abs(d) + _shift_rows(g.T, d, cap).T)
    top = float(cap if mask0.any() else 0)  # mask.max() (:98): 0 / 0 = NaN if nothing is set
    with np.errstate(invalid="ignore"):
        return (cap - dist).astype(float) / top
