"""``numpy.random.RandomState.randn`` streams on the GPU (csrc/rng.hip).

The STEPS member loop owns one ``RandomState`` per ensemble member (pysteps/nowcasts/steps.py:885-898)
and draws ``randstate.randn(m, n)`` from it once per time step (pysteps/noise/fftgenerators.py:400).
:class:`DeviceRandomStates` takes over a list of such generators - their ``get_state()`` is the whole
hand-over - and continues all of their streams on the device: MT19937 words, the polar method's
accept / reject decisions and the generators' final states are bit-identical with NumPy's, the values
are wherever the C library's ``log`` is correctly rounded (1 ulp otherwise, csrc/cr_log.h).
:meth:`DeviceRandomStates.sync_back` writes the states back into the host generators, so whatever the
caller draws afterwards continues the same stream.
"""

import ctypes

import numpy as np

from .. import _lib
from ..device import DeviceArray

__all__ = ["DeviceRandomStates"]


class DeviceRandomStates:
    """``len(randstates)`` legacy MT19937 generators continued on the device.

    ``max_draw``: the largest number of values one :meth:`randn` call will ask for per generator.
    ``n_draws``: how many such draws are expected (0 / None: unknown).  With a hint the streams are cut
    into chunks whose start states come from MT19937's jump-ahead polynomials, so that a draw is
    produced by hundreds of workgroups instead of one per generator; draws beyond the hint still work
    (one workgroup per generator)."""

    def __init__(self, randstates, max_draw, n_draws=None):
        self._lib = _lib.lib()
        self.randstates = list(randstates)
        if not self.randstates:
            raise ValueError("at least one RandomState is required")
        keys, pos, has, cached = [], [], [], []
        for rs in self.randstates:
            st = rs.get_state(legacy=True)
            if st[0] != "MT19937":
                raise ValueError("only MT19937 generators (numpy.random.RandomState) can be continued on the device")
            keys.append(np.asarray(st[1], dtype=np.uint32))
            pos.append(int(st[2]))
            has.append(int(st[3]))
            cached.append(float(st[4]))
        self.n = len(self.randstates)
        self.max_draw = int(max_draw)
        k = np.ascontiguousarray(np.stack(keys), dtype=np.uint32)
        p = np.asarray(pos, dtype=np.int32)
        h = np.asarray(has, dtype=np.int32)
        c = np.asarray(cached, dtype=np.float64)
        handle = ctypes.c_void_p()
        _lib.check(self._lib.psh_rng_create(self.n, k.ctypes.data, p.ctypes.data, h.ctypes.data, c.ctypes.data,
                                            self.max_draw, int(min(n_draws or 0, 1 << 20)), ctypes.byref(handle)), "psh_rng_create")
        self._h = handle

    def randn(self, *shape, out=None, side=False):
        """The next ``prod(shape)`` standard normal values of every generator: a float64 DeviceArray
        ``(n_generators, *shape)``, row ``j`` = ``randstates[j].randn(*shape)``.  ``side=True`` queues
        the draw on the generators' own stream behind everything queued so far; call :meth:`wait`
        before the result is used (or freed)."""
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if out is None:
            out = DeviceArray((self.n,) + tuple(int(s) for s in shape), np.float64)
        elif out.dtype != np.float64 or out.size != self.n * count:
            raise ValueError("out must be a float64 DeviceArray of n_generators * prod(shape) values")
        _lib.check(self._lib.psh_rng_randn_dev(self._h, count, out.ptr, 1 if side else 0), "psh_rng_randn_dev")
        return out

    def uniform(self, low, high, *shape, out=None, side=False):
        """The next ``prod(shape)`` values of ``uniform(low, high)`` of every generator (two words per value, like
        ``RandomState.uniform``): a float64 DeviceArray ``(n_generators, *shape)``."""
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if out is None:
            out = DeviceArray((self.n,) + tuple(int(s) for s in shape), np.float64)
        elif out.dtype != np.float64 or out.size != self.n * count:
            raise ValueError("out must be a float64 DeviceArray of n_generators * prod(shape) values")
        _lib.check(self._lib.psh_rng_uniform_dev(self._h, count, float(low), float(high), out.ptr, 1 if side else 0),
                   "psh_rng_uniform_dev")
        return out

    def wait(self):
        """Make the library stream wait for a ``side=True`` draw."""
        _lib.check(self._lib.psh_rng_wait(self._h), "psh_rng_wait")

    def check(self):
        """Raise when a draw that has completed came out short (no device work, see ``psh_rng_check``)."""
        _lib.check(self._lib.psh_rng_check(self._h), "psh_rng_check")

    def get_states(self):
        """The generators' states in ``RandomState.get_state()`` form (waits for the queued draws)."""
        k = np.empty((self.n, 624), dtype=np.uint32)
        p = np.empty(self.n, dtype=np.int32)
        h = np.empty(self.n, dtype=np.int32)
        c = np.empty(self.n, dtype=np.float64)
        _lib.check(self._lib.psh_rng_get_state(self._h, k.ctypes.data, p.ctypes.data, h.ctypes.data, c.ctypes.data),
                   "psh_rng_get_state")
        return [("MT19937", k[j].copy(), int(p[j]), int(h[j]), float(c[j])) for j in range(self.n)]

    def sync_back(self):
        """Write the device states back into the host generators handed to the constructor."""
        for rs, st in zip(self.randstates, self.get_states()):
            rs.set_state(st)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.psh_rng_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
