"""Lucas-Kanade against OpenCV itself - switched on by tests/golden/lk_opencv.npz, which
tools/make_lk_golden_cv2.py writes on any machine with cv2 + pysteps (none of the boxes this
repository was built on has OpenCV: profiles/r0*/a_cv2_probe.txt).  Until that file exists these tests
skip and LK parity stays "unpinned" (DESIGN.md section 4); with it

* CPU: the restatement oracle/lk_opencv.py is pinned - opening, uint8 renderings and the corner list
  bit for bit, tracked vectors within 1e-2 px, same status;
* GPU: the device path against the same vectors, dense field rel-L2 <= 1e-3.
"""

import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2

PATH = os.path.join(GOLDEN, "lk_opencv.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="no OpenCV golden vectors (tools/make_lk_golden_cv2.py "
                                "has not been run on a box with cv2): LK parity unpinned")


@pytest.fixture(scope="module")
def golden():
    z = np.load(PATH, allow_pickle=False)
    return z, [str(c) for c in z["cases"]]


def _pairs(z, name):
    frames = z[name + "/frames"]
    return frames, np.isfinite(frames[0]), np.isfinite(frames[1])


def test_oracle_pinned_on_opencv(golden):
    from oracle import lk_opencv as olk

    z, cases = golden
    for name in cases:
        frames, pv, nv = _pairs(z, name)
        prev = olk.morph_opening(frames[0].astype(np.float64), pv, frames[0][pv].min())
        nxt = olk.morph_opening(frames[1].astype(np.float64), nv, frames[1][nv].min())
        np.testing.assert_array_equal(prev[pv], z[name + "/opened_prev"][pv])
        np.testing.assert_array_equal(nxt[nv], z[name + "/opened_next"][nv])
        lo, hi = prev[pv].min(), prev[pv].max()
        np.testing.assert_array_equal(olk.to_uint8(prev, pv, lo, hi, lo), z[name + "/prev_u8"])
        pts = olk.shitomasi_detection(prev, pv)
        np.testing.assert_array_equal(pts, z[name + "/corners"])
        xy, uv = olk.track_features(prev, nxt, pv, nv, pts)
        want_xy, want_uv = z[name + "/track_xy"], z[name + "/track_uv"]
        assert xy.shape == want_xy.shape
        np.testing.assert_array_equal(xy, want_xy)
        assert np.max(np.abs(uv - want_uv)) <= 1e-2
        dense = olk.dense_lucaskanade(frames.astype(np.float64))
        assert rel_l2(dense, z[name + "/dense"]) <= 1e-3


@pytest.mark.gpu
def test_device_path_against_opencv(golden):
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.motion import get_method
    from pysteps_amd.motion import lucaskanade as lkmod

    z, cases = golden
    dense_lk = get_method("LK")
    for name in cases:
        frames, pv, nv = _pairs(z, name)
        prep = lkmod.PreparedFrame(DeviceArray.from_host(frames[0].astype(np.float64)), 3, 5, True)
        nprep = lkmod.PreparedFrame(DeviceArray.from_host(frames[1].astype(np.float64)), 3, 5, False)
        clean = prep.clean.to_host()
        np.testing.assert_array_equal(clean[pv], z[name + "/opened_prev"][pv].astype(np.float32))
        np.testing.assert_array_equal(prep.track_u8.to_host(), z[name + "/prev_u8"])
        np.testing.assert_array_equal(nprep.track_u8.to_host(), z[name + "/next_u8"])
        np.testing.assert_array_equal(prep.feature_u8.to_host(), z[name + "/feature_u8"])
        pts = lkmod.detect_corners(prep)
        np.testing.assert_array_equal(pts, z[name + "/corners"])
        sxy, suv = dense_lk(frames.astype(np.float64), dense=False)
        want_xy, want_uv = z[name + "/sparse_xy"], z[name + "/sparse_uv"]
        assert sxy.shape == want_xy.shape
        np.testing.assert_array_equal(sxy, want_xy)
        assert np.max(np.abs(suv - want_uv)) <= 1e-2
        assert rel_l2(dense_lk(frames.astype(np.float64)), z[name + "/dense"]) <= 1e-3
