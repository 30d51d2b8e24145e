"""FFT method object for pysteps backed by the hand-written HIP transforms (csrc/fft.hip).

Mirror of the objects pysteps/utils/fft.py:20-37 (``get_numpy``) builds: a namespace with
``fft2, ifft2, rfft2, irfft2, fftshift, ifftshift, fftfreq`` (and ``fftn`` when ``fftn_shape`` is
given), which the noise generators (pysteps/noise/fftgenerators.py), the cascade decomposition
(pysteps/cascade/decomposition.py:136) and the STEPS member loop (pysteps/nowcasts/steps.py:637,
1008, 1111, 1189) call through ``fft_method``.  Everything is float64 / complex128 like numpy.

* NumPy in -> NumPy out: the array crosses the bus, is transformed on the GPU and comes back - a
  4096 x 4096 ``rfft2`` is then bound by PCIe (~5 ms) instead of pocketfft (~0.5 s).
* :class:`~pysteps_amd.device.DeviceArray` in -> DeviceArray out: nothing leaves HBM.
* Single-precision input gives single-precision output, as numpy >= 2.0 does (``rfft2`` of float32
  is complex64): the transform itself always runs in float64 and the result is rounded once.
* Any side length is taken: powers of two up to 8192 directly, every other length up to 4096
  through Bluestein's chirp-z identity inside the same kernels (radar composites are 640 x 710,
  1226 x 760, ...).  What is left - longer sides, anything but two dimensions - is handed to
  ``numpy.fft``, the reference's default method: same results, CPU speed.
"""

from types import SimpleNamespace

import numpy as np

from .. import _lib
from ..device import DeviceArray

MAX_SIDE = 8192  # powers of two
MAX_ANY_SIDE = 4096  # other lengths: the chirp-z transform needs 2 n - 1 <= 8192 points of LDS


def supported_shape(shape):
    """True if the HIP kernels transform this 2-d shape: each side a power of two in 2..8192 or any
    length in 2..4096."""
    def side(s):
        s = int(s)
        return 2 <= s <= MAX_ANY_SIDE or (s <= MAX_SIDE and s >= 2 and (s & (s - 1)) == 0)

    return len(shape) == 2 and all(side(s) for s in shape)


def _to_device(x, dtype):
    if isinstance(x, DeviceArray):
        if x.dtype != np.dtype(dtype):
            raise ValueError("device-resident FFT input must be %s (got %s)" % (np.dtype(dtype), x.dtype))
        return x, True
    return DeviceArray.from_host(np.ascontiguousarray(x, dtype=dtype)), False


def _finish(out, resident, single=False):
    if resident:
        return out
    host = out.to_host()
    if single:  # numpy >= 2.0 keeps single precision: float32 / complex64 in -> complex64 / float32 out
        return host.astype(np.complex64 if host.dtype == np.complex128 else np.float32)
    return host


def _is_single(x):
    return not isinstance(x, DeviceArray) and np.asarray(x).dtype in (np.float32, np.float16, np.complex64)


def rfft2(x):
    """numpy.fft.rfft2 of a real (m, n) array -> (m, n//2+1) complex128."""
    if not supported_shape(x.shape):
        return np.fft.rfft2(_host(x))
    if not isinstance(x, DeviceArray) and np.iscomplexobj(x):
        x = np.real(x)  # numpy discards the imaginary part (with a ComplexWarning)
    single = _is_single(x)
    d, resident = _to_device(x, np.float64)
    m, n = d.shape
    out = DeviceArray((m, n // 2 + 1), np.complex128)
    _lib.check(_lib.lib().psh_fft_rfft2_dev(d.ptr, m, n, out.ptr), "psh_fft_rfft2_dev")
    return _finish(out, resident, single)


def irfft2(x, s):
    """numpy.fft.irfft2(x, s=s) for x of shape (s[0], s[1]//2+1) -> real (s[0], s[1]) float64."""
    s = tuple(int(v) for v in s)
    if not supported_shape(s) or tuple(x.shape) != (s[0], s[1] // 2 + 1):
        return np.fft.irfft2(_host(x), s=s)
    single = _is_single(x)
    d, resident = _to_device(x, np.complex128)
    out = DeviceArray(s, np.float64)
    _lib.check(_lib.lib().psh_fft_irfft2_dev(d.ptr, s[0], s[1], out.ptr), "psh_fft_irfft2_dev")
    return _finish(out, resident, single)


def _c2c(x, inverse):
    if not supported_shape(x.shape):
        return (np.fft.ifft2 if inverse else np.fft.fft2)(_host(x))
    single = _is_single(x)
    d, resident = _to_device(x, np.complex128)
    m, n = d.shape
    out = DeviceArray((m, n), np.complex128)
    _lib.check(_lib.lib().psh_fft_c2c2_dev(d.ptr, m, n, 1 if inverse else 0, out.ptr), "psh_fft_c2c2_dev")
    return _finish(out, resident, single)


def fft2(x):
    """numpy.fft.fft2 of an (m, n) array -> complex128."""
    return _c2c(x, False)


def ifft2(x):
    """numpy.fft.ifft2 of an (m, n) array -> complex128."""
    return _c2c(x, True)


def _host(x):
    return x.to_host() if isinstance(x, DeviceArray) else x


def get_hip(shape, fftn_shape=None, **kwargs):
    """The FFT method object for fields of ``shape`` (signature of pysteps.utils.fft.get_numpy;
    ``n_threads`` and other keyword arguments of the CPU methods are accepted and ignored)."""
    shape = tuple(int(v) for v in shape)
    f = {
        "fft2": fft2,
        "ifft2": ifft2,
        "rfft2": rfft2,
        "irfft2": lambda X: irfft2(X, shape),
        # index shuffles / frequency vectors: host helpers of numpy, as in every reference method
        "fftshift": np.fft.fftshift,
        "ifftshift": np.fft.ifftshift,
        "fftfreq": np.fft.fftfreq,
    }
    if fftn_shape is not None:
        f["fftn"] = np.fft.fftn
    return SimpleNamespace(**f)
