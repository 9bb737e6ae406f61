"""conan_slam_amd -- MI355X-native (gfx950) EKF-SLAM / FastSLAM-2 hot path behind a C ABI.

The arithmetic lives in hand-written HIP kernels (conan_slam_amd/csrc) exported through
include/cslam.h as libcslam_hip.so; this package is the thin host-side mirror of the reference's
`Slam` call surface for that path.  There is no CPU fallback: using the classes without the built
library or without a GPU raises.
"""
from . import _capi  # noqa: F401
from ._capi import (CslamError, F32, F64, Q_LOWER_CHOL_GAIN, Q_PREDICT_NM4, Q_REF_EXACT, Q_TEXTBOOK,  # noqa: F401
                    device_count)
from .ekf import EKF, EKFBatch  # noqa: F401
from .sim import Simulator  # noqa: F401

__all__ = ["EKF", "EKFBatch", "Simulator", "CslamError", "F32", "F64", "Q_REF_EXACT", "Q_TEXTBOOK", "device_count"]
