"""Feature detectors of the dense Lucas-Kanade front end (reference: pysteps/feature/)."""
from .interface import get_method  # noqa: F401
