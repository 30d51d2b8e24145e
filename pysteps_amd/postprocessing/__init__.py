"""Mirrors of pysteps.postprocessing operators that sit inside the nowcast member loops."""

from . import probmatching  # noqa: F401
