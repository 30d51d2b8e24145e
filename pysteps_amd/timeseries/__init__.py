"""Mirrors of pysteps.timeseries operators that sit inside the nowcast member loops."""

from . import autoregression  # noqa: F401
