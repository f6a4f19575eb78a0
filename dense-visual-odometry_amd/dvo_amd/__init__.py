"""dvo_amd — MI355X-native dense RGB-D direct-alignment core behind the Python API of
pfontana96/dense-visual-odometry (`get_dvo(...)` / `DenseVisualOdometry.step()`).

Importing the package does not touch the GPU; constructing a backend does, and fails loudly
(`DVOHipError`) when the HIP extension or a device is missing — there is no CPU fallback.
"""
from .lie import Se3, So3                                   # noqa: F401
from .camera import RGBDCameraModel                          # noqa: F401
from .robust_dvo_hip import (RobustDVOHIP, get_dvo, robust_dvo_factory, DVOError,   # noqa: F401
                             ImagePyramidError)
from ._cabi import HipContext, DVOHipError                   # noqa: F401

__version__ = "0.1.0"
