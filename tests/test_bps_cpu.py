"""pysteps_amd.noise.motion.initialize_bps (unit fields shared between the members' perturbators)
against the reference's initialize_bps / generate_bps (pysteps/noise/motion.py:55-180, oracle/_ref)."""

import numpy as np
import pytest


def test_perturbator_identical_with_the_reference(ref_pysteps):
    from pysteps.noise.motion import generate_bps, initialize_bps as ref

    from pysteps_amd.noise.motion import initialize_bps

    V = np.random.default_rng(0).standard_normal((2, 50, 70))
    V[:, 3, 4] = 0.0  # |V| <= 1e-12: unit vector defined as zero (:130-131)
    for seed, kw in ((3, {}), (4, dict(p_par=(1.0, 0.5, 0.1), p_perp=(2.0, 0.2, -0.3)))):
        a = ref(V, 2.0, 5.0, randstate=np.random.RandomState(seed), **kw)
        b = initialize_bps(V, 2.0, 5.0, randstate=np.random.RandomState(seed), **kw)
        assert set(a) == set(b)
        for k in ("vsf", "p_par", "p_perp", "eps_par", "eps_perp"):
            assert a[k] == b[k], k
        np.testing.assert_array_equal(a["V_par"], b["V_par"])
        np.testing.assert_array_equal(a["V_perp"], b["V_perp"])
        np.testing.assert_array_equal(generate_bps(a, 15.0), generate_bps(b, 15.0))
    c = initialize_bps(V, 2.0, 5.0, randstate=np.random.RandomState(9))
    assert c["V_par"] is b["V_par"] and not c["V_par"].flags.writeable  # shared, read-only
    V[0, 0, 0] += 1.0  # the motion field changed in place: the shared copy must not be reused
    d = initialize_bps(V, 2.0, 5.0, randstate=np.random.RandomState(9))
    np.testing.assert_array_equal(d["V_par"], ref(V, 2.0, 5.0, randstate=np.random.RandomState(9))["V_par"])
    with pytest.raises(ValueError):
        initialize_bps(V[0], 2.0, 5.0)
    with pytest.raises(ValueError):
        initialize_bps(V, 2.0, 5.0, p_par=(1.0, 2.0))


def test_registered_as_a_velocity_perturbation_method(ref_pysteps):
    from pysteps import noise
    from pysteps.noise.motion import generate_bps

    from pysteps_amd import register
    from pysteps_amd.noise.motion import initialize_bps

    try:
        register.register()
    except Exception as exc:  # no GPU here: only the table entries are of interest
        pytest.skip(str(exc))
    init, gen = noise.get_method("bps_hip")
    assert init is initialize_bps and gen is generate_bps
