"""Motion-field perturbator of the STEPS ensemble, mirror of ``pysteps.noise.motion.initialize_bps``
(pysteps/noise/motion.py:55-143) for the member loop of nowcasts.steps (steps.py:904-936).

The reference builds the unit fields ``V_par = V / |V|`` and ``V_perp`` again for every ensemble
member - the same motion field every time: 0.5 s per member at 4096 x 4096, 3 s of the nowcaster's
initialisation with six members - and keeps a private copy per member (268 MB each).  What differs
between members is two Laplace variates.  Here the unit fields are computed once per motion field and
shared (read-only) between the members' perturbator dictionaries; the dictionaries have the
reference's keys, so ``generate_bps`` (noise/motion.py:146-180), the reference's loop and the
member-batched kernel (which only needs the scalars) take them unchanged.
``register()`` adds the pair as ``vel_pert_method="bps_hip"``.
"""

import weakref

import numpy as np

__all__ = ["initialize_bps", "get_default_params_bps_par", "get_default_params_bps_perp"]

_unit_cache = {}


def get_default_params_bps_par():
    """Default parameters of the parallel component (noise/motion.py:43-47, BPS2006)."""
    return (10.88, 0.23, -7.68)


def get_default_params_bps_perp():
    """Default parameters of the perpendicular component (noise/motion.py:49-52, BPS2006)."""
    return (5.76, 0.31, -2.72)


def _fingerprint(V):
    sample = V[:, :: max(1, V.shape[1] // 61), :: max(1, V.shape[2] // 67)]
    return (V.shape, str(V.dtype), float(np.sum(sample, dtype=np.float64)), float(V[0, 0, 0]), float(V[1, -1, -1]))


def _unit_fields(V):
    """(V_par, V_perp) of noise/motion.py:127-140, shared between calls with the same motion field."""
    key = id(V)
    hit = _unit_cache.get(key)
    fp = _fingerprint(V)
    if hit is not None and hit[0]() is V and hit[1] == fp:
        return hit[2], hit[3]
    N = np.linalg.norm(V, axis=0)
    mask = N > 1e-12
    V_n = np.zeros(V.shape)
    np.divide(V, N, out=V_n, where=mask)  # V[:, mask] / N[mask]; 0 elsewhere (:130-131)
    V_perp = np.stack([-V_n[1, :, :], V_n[0, :, :]])
    V_n.flags.writeable = False
    V_perp.flags.writeable = False
    try:
        ref = weakref.ref(V, lambda _r, k=key: _unit_cache.pop(k, None))
        _unit_cache[key] = (ref, fp, V_n, V_perp)
    except TypeError:
        pass
    return V_n, V_perp


def initialize_bps(V, pixelsperkm, timestep, p_par=None, p_perp=None, randstate=None, seed=None):
    """Initialize the motion field perturbator of BPS2006 (parameters, return value and errors as
    documented for the reference, noise/motion.py:58-96); ``V_par`` / ``V_perp`` of the returned
    dictionary are read-only arrays shared with the other perturbators of the same motion field."""
    if len(V.shape) != 3:
        raise ValueError("V is not a three-dimensional array")
    if V.shape[0] != 2:
        raise ValueError("the first dimension of V is not 2")
    if p_par is None:
        p_par = get_default_params_bps_par()
    if p_perp is None:
        p_perp = get_default_params_bps_perp()
    if len(p_par) != 3:
        raise ValueError("the length of p_par is not 3")
    if len(p_perp) != 3:
        raise ValueError("the length of p_perp is not 3")
    if randstate is None:
        randstate = np.random
    if seed is not None:
        randstate.seed(seed)
    eps_par = randstate.laplace(scale=1.0 / np.sqrt(2))
    eps_perp = randstate.laplace(scale=1.0 / np.sqrt(2))
    vsf = 60.0 / (timestep * pixelsperkm)  # advection velocities -> km/h (:124)
    V_par, V_perp = _unit_fields(V)
    return {"randstate": randstate, "vsf": vsf, "p_par": p_par, "p_perp": p_perp, "eps_par": eps_par,
            "eps_perp": eps_perp, "V_par": V_par, "V_perp": V_perp}
