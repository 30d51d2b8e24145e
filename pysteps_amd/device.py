"""Device-resident arrays for the HIP path (thin, explicit; no tensor library).

A :class:`DeviceArray` owns (or views) a ``psh_malloc`` allocation and knows its
shape/dtype so that fields can stay in HBM between the LK estimate, the
extrapolator and repeated nowcast steps.
"""

import ctypes

import numpy as np

from . import _lib


class DeviceArray:
    # uv_pairs: for a (2, m, n) float32 motion field made by dense_lucaskanade, the same field as (m, n, 2) {u, v}
    # pairs (written by the interpolation kernel) - the layout the extrapolator gathers from, so that it need not
    # interleave the planes on every call.  Whoever writes into the array's memory through `ptr` sets it to None;
    # the handles a writer would go through drop it themselves: view() (a child array's writes are invisible to
    # the parent), fill_bytes() and free().
    __slots__ = ("ptr", "shape", "dtype", "_owner", "_keep", "uv_pairs", "__weakref__")

    def __init__(self, shape, dtype=np.float32, ptr=None, owner=None):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self._keep = None  # objects that must outlive work queued on this array
        self.uv_pairs = None
        if ptr is None:
            p = ctypes.c_void_p()
            _lib.check(_lib.lib().psh_malloc(ctypes.byref(p), self.nbytes), "psh_malloc")
            self.ptr = p.value or 0
            self._owner = True
        else:
            self.ptr = int(ptr)
            self._owner = owner  # keeps the parent allocation alive for views

    # -- properties -----------------------------------------------------
    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    # -- construction / transfer ---------------------------------------
    @classmethod
    def from_host(cls, array, dtype=None, sync=True):
        src = np.asarray(array)
        if dtype is not None and np.dtype(dtype) == np.float32 and src.dtype == np.float64 and src.size >= (1 << 16):
            # float64 -> float32 on the device: the array crosses the bus as it is, a host-side
            # astype of a 4096^2 plane costs more than its transfer
            raw = cls.from_host(src, None, sync=sync)
            out = cls(raw.shape, np.float32)
            _lib.check(_lib.lib().psh_convert_dev(raw.ptr, out.ptr, raw.size, 0), "psh_convert_dev")
            out._keep = raw._keep
            return out  # `raw` goes back to the stream-ordered block cache
        arr = np.ascontiguousarray(array, dtype=dtype)
        out = cls(arr.shape, arr.dtype)
        if arr.nbytes:
            _lib.check(_lib.lib().psh_memcpy_h2d(out.ptr, arr.ctypes.data, arr.nbytes), "h2d")
            if sync:
                # pageable source: make sure the copy is complete before `arr` can die
                _lib.check(_lib.lib().psh_sync(), "sync")
            else:
                out._keep = arr  # the host buffer lives as long as the device array
        return out

    def to_host(self, out=None, dtype=None):
        """Copy to a NumPy array.  ``dtype=np.float64`` widens a float32 array on the device first
        (the reference's operators return float64 where pysteps feeds them float64)."""
        if dtype is not None and np.dtype(dtype) != self.dtype:
            if self.dtype != np.float32 or np.dtype(dtype) != np.float64:
                return self.to_host().astype(dtype)
            wide = DeviceArray(self.shape, np.float64)
            _lib.check(_lib.lib().psh_convert_dev(self.ptr, wide.ptr, self.size, 1), "psh_convert_dev")
            return wide.to_host(out)
        if out is None:
            from . import _pinned  # noqa: PLC0415

            out = _pinned.empty(self.shape, self.dtype)
        elif out.shape != self.shape or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise ValueError("to_host: out must be C-contiguous with matching shape/dtype")
        if self.nbytes:
            _lib.check(_lib.lib().psh_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes), "d2h")
        return out

    def view(self, index):
        """Sub-array along the leading axis (no copy)."""
        if self.ndim < 2 or not (0 <= index < self.shape[0]):
            raise IndexError("view index out of range")
        stride = self.nbytes // self.shape[0]
        self.uv_pairs = None  # a plane handed out may be written through: the planes are the truth from here on
        return DeviceArray(self.shape[1:], self.dtype, ptr=self.ptr + index * stride, owner=self)

    def fill_bytes(self, byte_value=0):
        self.uv_pairs = None
        _lib.check(_lib.lib().psh_memset(self.ptr, int(byte_value), self.nbytes), "memset")
        return self

    def free(self):
        if self._owner is True and self.ptr:
            _lib.check(_lib.lib().psh_free(self.ptr), "psh_free")
        self.ptr = 0
        self._owner = None
        self.uv_pairs = None

    def __del__(self):
        try:
            if self._owner is True and self.ptr:
                _lib.load().psh_free(self.ptr)
        except Exception:
            pass

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, ptr=0x%x)" % (self.shape, self.dtype, self.ptr)


def synchronize():
    _lib.check(_lib.lib().psh_sync(), "psh_sync")


class Event:
    """HIP event on the library stream (kernel timing in bench.py)."""

    def __init__(self):
        p = ctypes.c_void_p()
        _lib.check(_lib.lib().psh_event_create(ctypes.byref(p)), "event_create")
        self._e = p.value

    def record(self):
        _lib.check(_lib.lib().psh_event_record(self._e), "event_record")
        return self

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        _lib.check(_lib.lib().psh_event_elapsed_ms(self._e, stop._e, ctypes.byref(ms)), "event_elapsed")
        return float(ms.value)

    def __del__(self):
        try:
            if self._e:
                _lib.load().psh_event_destroy(self._e)
        except Exception:
            pass


def device_info():
    lib = _lib.lib()
    dev, cus = ctypes.c_int(), ctypes.c_int()
    tot, free = ctypes.c_size_t(), ctypes.c_size_t()
    name = ctypes.create_string_buffer(256)
    _lib.check(
        lib.psh_device_info(ctypes.byref(dev), ctypes.byref(cus), ctypes.byref(tot), ctypes.byref(free), name, 256),
        "device_info",
    )
    return {
        "device": dev.value,
        "cu_count": cus.value,
        "hbm_total": tot.value,
        "hbm_free": free.value,
        "name": name.value.decode(),
    }
