"""Device-resident ensemble advection (member-batched semi-Lagrangian steps).

The generic nowcast loop of the reference advects every ensemble member once per
time step with its own (optionally perturbed) velocity and threads the
displacement through as state (pysteps/nowcasts/utils.py:441-462).
:class:`EnsembleAdvector` does that for all members in one kernel launch
(``psh_semilag_members_state_dev``): the trajectories of all members stay in HBM, the
BPS velocity perturbation (pysteps/noise/motion.py:146-180) is applied in-kernel
from two scalars per member.
"""

import numpy as np

from .. import _lib
from ..device import DeviceArray

__all__ = ["EnsembleAdvector", "bps_scalars", "steps_perturbators", "steps_noise_generators"]

# defaults of pysteps/noise/motion.py:43-52 (BPS2006)
_BPS_PAR, _BPS_PERP = (10.88, 0.23, -7.68), (5.76, 0.31, -2.72)


def steps_perturbators(n_members, seed, kmperpixel, timestep, p_par=None, p_perp=None):
    """The scalar part of the velocity perturbators ``nowcasts.steps`` builds for ``n_members``
    members from ``seed`` - recomputable on every rank without communication.

    Seed chain of pysteps/nowcasts/steps.py:885-898: per member ``rs = RandomState(seed)`` (precip
    noise), ``seed = rs.randint(0, 1e9)``, ``rs = RandomState(seed)`` (motion), ``seed =
    rs.randint(0, 1e9)``; ``initialize_bps`` (noise/motion.py:119-125) then draws ``eps_par`` and
    ``eps_perp`` from the motion generator's Laplace distribution (scale 1/sqrt(2)) and sets
    ``vsf = 60 / (timestep * pixelsperkm)`` with ``pixelsperkm = 1 / kmperpixel`` (steps.py:924-929).
    Returns one dict per member for :class:`EnsembleAdvector`."""
    out = []
    for _ in range(int(n_members)):
        rs = np.random.RandomState(seed)
        seed = rs.randint(0, high=int(1e9))
        rs = np.random.RandomState(seed)
        seed = rs.randint(0, high=int(1e9))
        # the randint above is drawn from the motion generator before initialize_bps uses it
        eps_par = rs.laplace(scale=1.0 / np.sqrt(2))
        eps_perp = rs.laplace(scale=1.0 / np.sqrt(2))
        out.append(dict(eps_par=float(eps_par), eps_perp=float(eps_perp),
                        p_par=tuple(p_par or _BPS_PAR), p_perp=tuple(p_perp or _BPS_PERP),
                        vsf=60.0 / (float(timestep) * (1.0 / float(kmperpixel)))))
    return out


def steps_noise_generators(n_members, seed):
    """The precipitation-noise generators ``nowcasts.steps`` creates for ``n_members`` members from
    ``seed`` (the same chain, pysteps/nowcasts/steps.py:885-898: the first ``RandomState`` of every
    member) - recomputable on every rank, so a rank that owns members ``a .. b`` takes ``[a:b]``."""
    out = []
    for _ in range(int(n_members)):
        rs = np.random.RandomState(seed)
        out.append(rs)
        seed = rs.randint(0, high=int(1e9))
        seed = np.random.RandomState(seed).randint(0, high=int(1e9))
    return out


def bps_scalars(perturbators, t):
    """Per-member scalars of ``generate_bps(perturbator, t)`` (noise/motion.py:146-180):
    the perturbation field is ``par * V_par + perp * V_perp``."""
    par, perp = [], []
    for p in perturbators:
        g_par = p["p_par"][0] * pow(t, p["p_par"][1]) + p["p_par"][2]
        g_perp = p["p_perp"][0] * pow(t, p["p_perp"][1]) + p["p_perp"][2]
        par.append(g_par * p["eps_par"] / p["vsf"])
        perp.append(g_perp * p["eps_perp"] / p["vsf"])
    return np.asarray(par, dtype=np.float64), np.asarray(perp, dtype=np.float64)


class EnsembleAdvector:
    """Stateful advection of ``n_members`` fields with a common motion field.

    ``perturbators``: optional list (one per member) of dicts with the scalar entries
    of ``pysteps.noise.motion.initialize_bps`` (``eps_par``, ``eps_perp``, ``p_par``,
    ``p_perp``, ``vsf``); the unit fields ``V_par``/``V_perp`` are rebuilt on the device.
    """

    def __init__(self, velocity, n_members, perturbators=None, n_iter=1, interp_order=1, outval=np.nan,
                 packed=True):
        self._lib = _lib.lib()
        self.velocity = velocity if isinstance(velocity, DeviceArray) else DeviceArray.from_host(velocity, np.float32)
        if self.velocity.ndim != 3 or self.velocity.shape[0] != 2 or self.velocity.dtype != np.float32:
            raise ValueError("velocity must be a (2,m,n) float32 field")
        self.n_members = int(n_members)
        self.m, self.n = self.velocity.shape[1:]
        if perturbators is not None and len(perturbators) != self.n_members:
            raise ValueError("one perturbator per member is required")
        self.perturbators = perturbators
        self.n_iter, self.interp_order, self.outval = int(n_iter), int(interp_order), float(outval)
        # trajectories between calls, in the kernel's own representation: one 16-byte record per
        # member and pixel {int32 P-x, int32 P-y, float32 frac_x, float32 frac_y} (half the HBM
        # traffic of the float64 displacement pair; see ``displacement``)
        self._state = DeviceArray((self.n_members, self.m, self.n, 4), np.uint32)
        self._started = False
        self.vhat = None
        if perturbators is not None:
            self.vhat = DeviceArray((2, self.m, self.n), np.float32)
            _lib.check(self._lib.psh_velocity_unit_dev(self.velocity.ptr, self.m, self.n, self.vhat.ptr),
                       "psh_velocity_unit_dev")
        # interleaved gather plane, built once per motion field: {u,v,V_par_x,V_par_y} or {u,v}
        self.packed = None
        if packed and self.m * self.n < (1 << 28):
            self.packed = DeviceArray((self.m, self.n, 4 if self.vhat is not None else 2), np.float32)
            _lib.check(self._lib.psh_members_pack_dev(self.velocity.ptr, None if self.vhat is None else self.vhat.ptr,
                                                      self.m, self.n, self.packed.ptr), "psh_members_pack_dev")

    def reset(self):
        """Back to lead time zero (all displacements zero) without releasing the resident state."""
        self._started = False

    @property
    def displacement(self):
        """The displacements ``D_j`` of all members as the reference carries them: a float64
        DeviceArray ``(n_members, 2, m, n)`` (zeros before the first step), converted from the
        resident trajectory records on demand."""
        disp = DeviceArray((self.n_members, 2, self.m, self.n), np.float64)
        if not self._started:
            return disp.fill_bytes(0)
        _lib.check(self._lib.psh_members_state_to_disp_dev(self._state.ptr, self.n_members, self.m, self.n,
                                                           disp.ptr), "psh_members_state_to_disp_dev")
        return disp

    @displacement.setter
    def displacement(self, value):
        """Restart from given displacements ((n_members, 2, m, n) float64, host or device)."""
        disp = value if isinstance(value, DeviceArray) else DeviceArray.from_host(value, np.float64)
        if disp.shape != (self.n_members, 2, self.m, self.n) or disp.dtype != np.float64:
            raise ValueError("displacement must have shape (n_members, 2, m, n) and dtype float64")
        _lib.check(self._lib.psh_members_disp_to_state_dev(disp.ptr, self.n_members, self.m, self.n,
                                                           self._state.ptr), "psh_members_disp_to_state_dev")
        self._started = True

    def step(self, precip_members, t_diff, t_total=None, out_dtype=None):
        """Advect all members by ``t_diff`` (lead-time increment(s) in velocity time steps, i.e.
        the ``timestep_diff`` of the reference - a sequence gives several steps in one call);
        ``t_total`` is the lead time
        handed to the perturbators (minutes, as in the reference).  ``precip_members``:
        (B,m,n) ndarray or float32 DeviceArray, or None for displacement only.
        Returns the advected members (B,m,n) in the container type of the input; host results in
        ``out_dtype`` (float32 unless given; float64 is widened on the device)."""
        on_device = isinstance(precip_members, DeviceArray)
        pm = None
        if precip_members is not None:
            pm = precip_members if on_device else DeviceArray.from_host(precip_members, np.float32)
            if pm.shape != (self.n_members, self.m, self.n) or pm.dtype != np.float32:
                raise ValueError("precip_members must have shape (n_members, m, n)")
        steps = np.atleast_1d(np.asarray(t_diff, dtype=np.float64))
        par = perp = None
        if self.perturbators is not None:
            if t_total is None:
                raise ValueError("t_total is required with velocity perturbations")
            par, perp = bps_scalars(self.perturbators, t_total)
        out = None if pm is None else DeviceArray((self.n_members, steps.size, self.m, self.n), np.float32)
        tail = (None if par is None else par.ctypes.data, None if perp is None else perp.ctypes.data,
                self.n_members, self.m, self.n, steps.ctypes.data, int(steps.size), self.n_iter,
                self.interp_order, self.outval, self._state.ptr, int(self._started),
                None if out is None else out.ptr)
        head = (None if pm is None else pm.ptr, self.velocity.ptr, None if self.vhat is None else self.vhat.ptr)
        if self.packed is not None:
            rc = self._lib.psh_semilag_members_packed_dev(*head, self.packed.ptr, *tail)
        else:
            rc = self._lib.psh_semilag_members_state_dev(*head, *tail)
        _lib.check(rc, "psh_semilag_members_state_dev")
        self._started = True
        if out is None:
            return None
        if steps.size == 1:
            out = DeviceArray((self.n_members, self.m, self.n), np.float32, ptr=out.ptr, owner=out)
        return out if on_device else out.to_host(dtype=out_dtype)
