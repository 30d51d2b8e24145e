"""``extrapolation.get_method`` mirror (reference: pysteps/extrapolation/interface.py:107-145).

Names resolve like in the reference - case-insensitive strings, ``None``/"none"
for the no-op, "eulerian" for persistence, anything else ``ValueError`` - with
"semilagrangian" served by the HIP kernel.
"""

import numpy as np

from .._registry import MethodTable
from . import semilagrangian


def eulerian_persistence(precip, velocity, timesteps, outval=np.nan, **kwargs):
    """Eulerian persistence: the input repeated once per lead time (reference :41-93).

    Like the reference, the optional displacement is an all-zero array shaped
    ``(2,) + out.shape``.
    """
    count = timesteps if isinstance(timesteps, int) else len(timesteps)
    frames = np.repeat(np.asarray(precip)[None, ...], count, axis=0)
    if kwargs.get("return_displacement", False):
        return frames, np.zeros((2,) + frames.shape)
    return frames


def _no_extrapolation(precip, velocity, timesteps, outval=np.nan, **kwargs):
    """``get_method(None)``: accepts the extrapolator arguments, returns None (reference :96-104)."""
    return None


_table = MethodTable("extrapolation")
_table.add("eulerian", eulerian_persistence)
_table.add(["semilagrangian", "semilagrangian_hip"], semilagrangian.extrapolate)
_table.add([None, "none"], _no_extrapolation)


def get_method(name):
    """Return the extrapolator registered under ``name`` (contract of reference :114-145)."""
    return _table.lookup(name)
