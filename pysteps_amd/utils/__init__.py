"""Sparse-vector utilities on the dense-LK path (mirrors of pysteps.utils.cleansing / interpolate)."""

from .cleansing import decluster, detect_outliers, detect_outliers_device  # noqa: F401
from .interpolate import idwinterp2d, rbfinterp2d  # noqa: F401
from .transformation import dB_transform  # noqa: F401,E402
from .check_norain import check_norain  # noqa: F401,E402
