"""HIP implementations behind pysteps' motion (optical flow) interface."""

from .interface import get_method  # noqa: F401
