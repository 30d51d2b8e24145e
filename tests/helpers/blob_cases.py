"""Cases of tests/golden/blob_reference.npz (written by the unmodified reference: tools/make_golden_blob.py)."""
import json
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "blob_reference.npz")


def load():
    g = np.load(PATH, allow_pickle=False)
    cases = []
    for name in g["names"]:
        name = str(name)
        q = g[name + "__image_q64"]
        image = np.where(q == -32768, np.nan, q / 64.0).astype(str(g[name + "__dtype"]))  # counts of 1/64, NaN = -32768
        cases.append((name, image, json.loads(str(g[name + "__kwargs"])), g[name + "__points"]))
    return cases, json.loads(str(g["versions"]))


def same_blobs(got, want):
    """Same array up to the last digits of the sigma column (np.logspace differs in the 15th digit between NumPy versions)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if got.shape != want.shape:
        return False
    if got.size == 0:
        return True
    return np.array_equal(got[:, :2], want[:, :2]) and np.allclose(got[:, 2:], want[:, 2:], rtol=1e-13, atol=0.0)
