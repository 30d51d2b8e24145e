"""Noise generators on the HIP path (mirror of pysteps.noise.fftgenerators)."""

from .fftgenerators import generate_noise_2d_fft_filter  # noqa: F401
