"""``motion.get_method`` mirror (reference: pysteps/motion/interface.py:36-111).

"lk"/"lucaskanade" resolve to the HIP dense Lucas-Kanade; ``None`` returns the
reference's zero-motion callable; the other reference methods (vet, darts,
proesmans, constant, farneback) are different algorithms outside this package and
are forwarded to pysteps when it is importable.
"""

import numpy as np

from .._registry import MethodTable
from .lucaskanade import dense_lucaskanade

_OTHER_REFERENCE_METHODS = ("vet", "darts", "proesmans", "constant", "farneback")


def _zero_motion(precip, *args, **kwargs):
    """``get_method(None)``: zero motion field for the last two axes of the input (reference :92-95)."""
    return np.zeros((2,) + tuple(precip.shape[-2:]))


_table = MethodTable("optical flow")
_table.add(["lk", "lucaskanade", "lk_hip", "lucaskanade_hip"], dense_lucaskanade)
_table.add(None, _zero_motion)


def get_method(name):
    """Return the optical-flow callable registered under ``name`` (contract of reference :49-111)."""
    if isinstance(name, str) and name.lower() in _OTHER_REFERENCE_METHODS:
        try:
            from pysteps.motion.interface import get_method as ref_get  # noqa: PLC0415
        except Exception as exc:
            raise NotImplementedError(
                "optical flow method %r is not part of pysteps_amd and pysteps is not importable" % name
            ) from exc
        return ref_get(name)
    return _table.lookup(name)
