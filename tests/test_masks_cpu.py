"""Incremental mask without a GPU: the distance-transform restatement (oracle/masks.py, the form
csrc/mask.hip evaluates) against the reference's compute_dilated_mask (pysteps/nowcasts/utils.py:69-101,
scipy.ndimage underneath) run live from oracle/_ref; the mirror's delegation and the registration hook."""

import warnings

import numpy as np
import pytest

from oracle import masks as oracle


def _structures():
    from scipy.ndimage import generate_binary_structure, iterate_structure

    cross = generate_binary_structure(2, 1)
    return [cross, iterate_structure(cross, 2), iterate_structure(cross, 3), np.ones((3, 3), bool),
            np.array([[1, 0, 0], [0, 1, 1], [0, 0, 0]], bool), np.ones((2, 4), bool),
            np.array([[0, 1, 1, 0, 1]], bool)]


def test_oracle_is_the_reference_bit_for_bit(ref_pysteps):
    from pysteps.nowcasts.utils import compute_dilated_mask as ref

    rng = np.random.default_rng(0)
    structures = _structures()
    for it in range(140):
        shape = (int(rng.integers(1, 40)), int(rng.integers(1, 40)))
        mask = rng.random(shape) < rng.choice([0.0, 0.01, 0.05, 0.3])
        kr, r = structures[it % len(structures)], int(rng.choice([0, 1, 3, 10]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # 0 / 0 of an empty mask
            want = ref(mask, kr, r)
        assert np.array_equal(oracle.compute_dilated_mask(mask, kr, r), want, equal_nan=True), (it, shape, r)
    values = rng.random((30, 30)) * 3  # not boolean: cast to uint8 first (:87), 0.5 -> 0
    assert np.array_equal(oracle.compute_dilated_mask(values, structures[0], 4), ref(values, structures[0], 4))


def test_small_and_odd_masks_are_the_reference(ref_pysteps):
    from pysteps.nowcasts.utils import compute_dilated_mask as ref

    from pysteps_amd.nowcasts.utils import compute_dilated_mask

    rng = np.random.default_rng(1)
    cross = _structures()[0]
    small = rng.random((40, 50)) < 0.1
    assert np.array_equal(compute_dilated_mask(small, cross, 10), ref(small, cross, 10))
    cube = rng.random((4, 70, 80)) < 0.1  # not two-dimensional: scipy's own answer / error
    with pytest.raises(Exception) as a:
        ref(cube, cross, 2)
    with pytest.raises(type(a.value)):
        compute_dilated_mask(cube, cross, 2)


def test_patch_and_unpatch(ref_pysteps):
    import pysteps.nowcasts.utils as ref_mod

    from pysteps_amd import register
    from pysteps_amd.nowcasts import utils as hip_mod

    stock = ref_mod.compute_dilated_mask
    try:
        assert register.patch_dilated_mask() == ["nowcasts.utils:compute_dilated_mask"]
        assert ref_mod.compute_dilated_mask is hip_mod.compute_dilated_mask
        assert register.patch_dilated_mask() == []
        assert hip_mod._reference_compute_dilated_mask() is stock
    finally:
        register.unpatch_dilated_mask()
    assert ref_mod.compute_dilated_mask is stock and not hasattr(ref_mod, "_reference_compute_dilated_mask")
