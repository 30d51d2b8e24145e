"""NumPy arrays on pinned host memory from the library's cached pool (``psh_host_alloc``).

The host-buffer entry points are transfer bound (a 4096^2 x 24 nowcast: 1.4 ms of kernel, 1.5 GiB
of results); a device-to-host copy into pinned memory runs at the speed of the link and needs no
staging pass.  ``empty()`` therefore hands out result arrays whose storage is a pinned block; the
block goes back to the pool when the last array referring to it is garbage collected.  The arrays
are ordinary, writable ``numpy.ndarray`` objects owned by the caller (the reference's ownership
rule: outputs are fresh arrays).  If the pool is exhausted (``PYSTEPS_HIP_PINNED_BYTES``, default
16 GiB) or the request is small, ordinary ``np.empty`` memory is returned instead.
"""

import ctypes

import numpy as np

from . import _lib

_MIN_BYTES = 1 << 20  # below this the staging ring is as fast and pinning is not worth a block


class _Block:
    """Owner of one pinned block; ``np.asarray(block)`` views it without copying."""

    __slots__ = ("ptr", "nbytes", "__array_interface__")

    def __init__(self, ptr, nbytes):
        self.ptr = ptr
        self.nbytes = nbytes
        self.__array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            _lib.load().psh_host_free(self.ptr)
        except Exception:
            pass


def empty(shape, dtype):
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in np.atleast_1d(shape))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < _MIN_BYTES:
        return np.empty(shape, dtype=dtype)
    p = ctypes.c_void_p()
    rc = _lib.lib().psh_host_alloc(ctypes.byref(p), nbytes)
    if rc != 0 or not p.value:
        return np.empty(shape, dtype=dtype)  # pool exhausted: the staged path takes over
    return np.asarray(_Block(p.value, nbytes)).view(dtype).reshape(shape)
