"""HIP implementations behind pysteps' extrapolation interface."""

from .interface import get_method  # noqa: F401
