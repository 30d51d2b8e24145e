"""Cascade decomposition on the HIP path (mirror of pysteps.cascade.decomposition)."""

from .decomposition import decomposition_fft, recompose_fft  # noqa: F401
