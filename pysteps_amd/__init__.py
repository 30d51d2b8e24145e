"""pysteps_amd - MI355X-native advection hot path for pysteps.

Dense Lucas-Kanade optical flow (``motion.get_method("LK")``) and semi-Lagrangian
extrapolation (``extrapolation.get_method("semilagrangian")``) as hand-written
HIP kernels for gfx950 behind a C ABI (``include/pysteps_hip.h``), with Python
shims that mirror the reference operators.  ``register()`` plugs them into
pysteps' own method tables.
"""

__version__ = "0.1.0"

from . import extrapolation, motion  # noqa: F401,E402
