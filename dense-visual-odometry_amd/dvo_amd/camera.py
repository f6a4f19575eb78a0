"""Pinhole RGB-D camera model — host-side mirror of the reference's `RGBDCameraModel`
(camera_model.py:17-256): same constructor, `at(level)`, `deproject`, `project`, `load_from_yaml`,
`intrinsics`, `depth_scale`.

The HIP kernels do their own back-projection / projection on the device (csrc/dvo_kernels.hip,
`dvo_geo`); the NumPy `deproject` / `project` here exist for API parity (tools and tests that call
them directly) and spell out the float32 operation order the kernels use.  There is no np_cache
(utils/numpy_cache.py): device-resident pyramids make the memoisation moot.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Union

import numpy as np
import yaml

logger = logging.getLogger(__name__)


class RGBDCameraModel:
    INTRINSICS_KEYWORD = "intrinsics"
    DEPTH_SCALE_KEYWORD = "depth_scale"
    DISTORSSION_COEFFS_KEYWORD = "distorssion_coefficients"
    DISTORSSION_MODEL_KEYWORD = "distorssion_model"

    def __init__(self, intrinsics: np.ndarray, depth_scale: float,
                 distorssion_coeffs: Union[np.ndarray, None] = None, distorssion_model: Union[str, None] = None):
        assert intrinsics.shape == (3, 3), f"Expected a 3x3 'intrinsics', got {intrinsics.shape} instead"
        assert depth_scale >= 0.0, \
            "Expected 'scale' to be a positive floating point, got '{:.3f}' instead".format(depth_scale)
        self._intrinsics = np.zeros((3, 4), dtype=np.float32)      # stored 3x4 like the reference (:55-57)
        self._intrinsics[:, :3] = intrinsics
        self.depth_scale = depth_scale
        self.distorssion_coeffs = distorssion_coeffs
        self.distorsion_model = distorssion_model

    @property
    def intrinsics(self) -> np.ndarray:
        return self._intrinsics

    def at(self, level: int) -> np.ndarray:
        """Level intrinsics: f * 2^-L, c * 2^-L + 2^-(L+1) - 0.5 (camera_model.py:62-79)."""
        assert level >= 0, "Expected 'level' to be >= 0, got '{}' instead".format(level)
        if level == 0:
            return self._intrinsics
        s = np.float32(2.0 ** -level)
        o = np.float32(2.0 ** (-level - 1) - 0.5)
        K = self._intrinsics
        out = np.zeros((3, 4), dtype=np.float32)
        out[0, 0], out[0, 1], out[0, 2] = s * K[0, 0], s * K[0, 1], s * K[0, 2] + o * K[2, 2]
        out[1, 1], out[1, 2] = s * K[1, 1], s * K[1, 2] + o * K[2, 2]
        out[1, 0] = s * K[1, 0]
        out[2, :3] = K[2, :3]
        return out

    @classmethod
    def load_from_yaml(cls, filepath: Path):
        """`intrinsics` (3x3) + `depth_scale` from YAML; None if missing/invalid (camera_model.py:81-114)."""
        if not filepath.exists():
            logger.error("Could not find configuration file '{}'".format(str(filepath)))
            return None
        with filepath.open("r") as fp:
            data = yaml.safe_load(fp)
        try:
            K = np.array(data[cls.INTRINSICS_KEYWORD], dtype=np.float32)
            scale = data[cls.DEPTH_SCALE_KEYWORD]
        except KeyError as e:
            logger.error(e)
            return None
        return cls(K, scale, data.get(cls.DISTORSSION_COEFFS_KEYWORD), data.get(cls.DISTORSSION_MODEL_KEYWORD))

    # -- geometry (host restatement of what the kernels do per pixel) ---------------------------------
    @staticmethod
    def _inverse_entries(KL: np.ndarray):
        f32 = np.float32
        fx, s, cx, fy, cy = f32(KL[0, 0]), f32(KL[0, 1]), f32(KL[0, 2]), f32(KL[1, 1]), f32(KL[1, 2])
        one = f32(1)
        ia, idd = one / fx, one / fy
        ie = (-cy) / fy
        ib = (-(s * idd)) / fx
        ic = ((-(s * ie)) - cx) / fx
        return ia, ib, ic, idd, ie

    def deproject(self, depth_image: np.ndarray, return_mask: bool = False, level: int = 0,
                  mask_pixels: np.ndarray = None):
        """Depth image -> 4xN homogeneous cloud of the pixels with depth != 0 (camera_model.py:171-226)."""
        h, w = depth_image.shape
        mask = depth_image != 0.0
        if mask_pixels is not None:
            mask = np.logical_and(mask, mask_pixels)
        flat = mask.reshape(-1)
        z = (depth_image.reshape(-1) * self.depth_scale)[flat].astype(np.float32)
        vv, uu = np.divmod(np.flatnonzero(flat), w)
        u = uu.astype(np.float32)
        v = vv.astype(np.float32)
        ia, ib, ic, idd, ie = self._inverse_entries(self.at(level))
        xn = ((ia * u) + (ib * v)) + ic
        yn = (idd * v) + ie
        cloud = np.vstack((xn * z, yn * z, z, np.ones_like(z)))
        if return_mask:
            return cloud, mask.reshape(depth_image.shape)
        return cloud

    def project(self, pointcloud: np.ndarray, level: int = 0) -> np.ndarray:
        """4xN cloud -> 3xN homogeneous pixel coordinates (camera_model.py:228-252)."""
        K = self.at(level)
        x, y, z = (np.asarray(pointcloud[i], dtype=np.float32) for i in range(3))
        with np.errstate(divide="ignore", invalid="ignore"):
            un = ((K[0, 0] * x) + (K[0, 1] * y)) + (K[0, 2] * z)
            vn = (K[1, 1] * y) + (K[1, 2] * z)
            return np.vstack((un / z, vn / z, z / z)).astype(np.float32)
