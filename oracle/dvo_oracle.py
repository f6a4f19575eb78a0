"""CPU restatement (NumPy) of the reference's dense RGB-D direct-alignment hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`dense-visual-odometry_amd/`)
may import this module; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
``cpu_baseline`` leg do, and only as the checker / reported CPU baseline.

Parity status: PINNED — checked in tests/test_oracle_golden.py against golden vectors produced
by executing the reference's own source files in the build container
(oracle/refrun/make_golden.py; per-iteration H, b, increments, dense r/J/mask, final poses)
and against the known-answer values of the reference's own tests.  Three OpenCV functions the
reference calls (cvtColor/medianBlur/Sobel; opencv-python ~=4.6, absent here) are restated from
OpenCV's published definitions: Sobel is pinned by the reference's 5x5 test, median and BGR2GRAY
values are "parity unpinned".

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/dense_visual_odometry/).

Arithmetic contract (this is what the HIP kernels reproduce bit-for-bit on the coordinate path):
all per-pixel geometry is IEEE float32, round-to-nearest, one rounding per operation, NO fused
multiply-add, in exactly the operation order written below.  The reference gets the same
quantities from float32 BLAS calls (np.dot), whose rounding order depends on the BLAS build and
CPU; the order fixed here is one valid realisation of it and is machine-independent.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Sequence, List, Optional, Tuple

import numpy as np
import scipy.linalg

F32 = np.float32
LIE_EPSILON = 1e-6          # utils/lie_algebra/base_special_group.py:8


# ================================================================================================
# OpenCV pieces (third-party; restated from the published algorithms)
# ================================================================================================
def bgr2gray(color: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(..., COLOR_BGR2GRAY) as called at core/base_dense_visual_odometry.py:58.
    OpenCV 4.x 8-bit path: (3735*B + 19235*G + 9798*R + 2^14) >> 15.  Parity unpinned."""
    c = np.asarray(color)
    assert c.dtype == np.uint8 and c.ndim == 3 and c.shape[2] == 3
    b = c[..., 0].astype(np.int32)
    g = c[..., 1].astype(np.int32)
    r = c[..., 2].astype(np.int32)
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def median3(img: np.ndarray) -> np.ndarray:
    """cv2.medianBlur(img, 3) (utils/image_pyramid.py:20): 3x3 median, replicated border."""
    a = np.asarray(img)
    h, w = a.shape
    p = np.pad(a, 1, mode="edge")
    stack = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], axis=0)
    return np.partition(stack, 4, axis=0)[4].astype(a.dtype)


def pyr_down_median(img: np.ndarray) -> np.ndarray:
    """pyrDownMedianSmooth, utils/image_pyramid.py:19-21: median then keep even rows/cols."""
    return np.ascontiguousarray(median3(img)[::2, ::2])


def build_pyramid(img: np.ndarray, levels: int) -> List[np.ndarray]:
    """ImagePyramid.__init__, utils/image_pyramid.py:35-54."""
    if levels < 0:
        raise ValueError("Expected 'levels' to be non-negative, got '{}' instead".format(levels))
    pyr = [None] * levels
    if levels:
        pyr[0] = np.asarray(img)
    for lvl in range(1, levels):
        pyr[lvl] = pyr_down_median(pyr[lvl - 1])
    return pyr


def sobel3(img: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """compute_gradients, utils/jacobian.py:47-73: cv2.Sobel(ksize=3, BORDER_REFLECT), float32,
    unnormalised ([-1 0 1] x [1 2 1]^T).  Pinned by tests/.../utils/test_jacobian.py:49-73."""
    a = np.pad(np.asarray(img).astype(np.float32), 1, mode="symmetric")
    h, w = np.asarray(img).shape

    def s(yy, xx):
        return a[yy:yy + h, xx:xx + w]
    gx = (s(0, 2) + 2 * s(1, 2) + s(2, 2)) - (s(0, 0) + 2 * s(1, 0) + s(2, 0))
    gy = (s(2, 0) + 2 * s(2, 1) + s(2, 2)) - (s(0, 0) + 2 * s(0, 1) + s(0, 2))
    return gx.astype(np.float32), gy.astype(np.float32)


def central_diff(img: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Device fn compute_gradients of the reference's numba.cuda variant (V3),
    cuda/residuals_kernel.py:8-19: 0.5*(I[x+1]-I[x-1]) with clamped indices."""
    a = np.asarray(img).astype(np.float32)
    p = np.pad(a, 1, mode="edge")
    gx = F32(0.5) * (p[1:-1, 2:] - p[1:-1, :-2])
    gy = F32(0.5) * (p[2:, 1:-1] - p[:-2, 1:-1])
    return gx.astype(np.float32), gy.astype(np.float32)


# ================================================================================================
# Camera model
# ================================================================================================
def level_intrinsics(K: np.ndarray, level: int) -> np.ndarray:
    """RGBDCameraModel.at, camera_model.py:62-79 -> 3x4 float32."""
    K = np.asarray(K, dtype=np.float32)
    out = np.zeros((3, 4), dtype=np.float32)
    if level == 0:
        out[:3, :3] = K[:3, :3]
        return out
    lv = -level
    scale = np.array([[2 ** lv, 0, 2 ** (lv - 1) - 0.5],
                      [0, 2 ** lv, 2 ** (lv - 1) - 0.5],
                      [0, 0, 1]], dtype=np.float32)
    out[:3, :3] = np.dot(scale, K[:3, :3])     # powers of two and one add per entry: exact in any BLAS order
    return out


def inverse_intrinsics(KL: np.ndarray) -> Tuple[F32, F32, F32, F32, F32]:
    """np.linalg.inv(intrinsics[:3,:3]) of camera_model.py:219, for the upper-triangular K the
    reference always has, spelled out as float32 back-substitution with true divisions:
        [[a b c], [0 d e], [0 0 1]],  a=1/fx, d=1/fy, e=-cy/fy, b=-(s*d)/fx, c=(-(s*e)-cx)/fx."""
    fx, s, cx = F32(KL[0, 0]), F32(KL[0, 1]), F32(KL[0, 2])
    fy, cy = F32(KL[1, 1]), F32(KL[1, 2])
    one = F32(1.0)
    a = one / fx
    d = one / fy
    e = (-cy) / fy
    b = (-(s * d)) / fx
    c = ((-(s * e)) - cx) / fx
    return F32(a), F32(b), F32(c), F32(d), F32(e)


def deproject(depth: np.ndarray, KL: np.ndarray, depth_scale: float):
    """RGBDCameraModel.deproject, camera_model.py:171-226.
    Returns x, y, z (float32, depth-valid pixels in row-major order) and mask (H x W bool)."""
    d = np.asarray(depth)
    h, w = d.shape
    mask = (d != 0)                                                   # :195
    zz = (d.reshape(-1).astype(np.float64) * float(depth_scale))      # :200 product in float64
    m = mask.reshape(-1)
    z = zz[m].astype(np.float32)                                      # :201
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))   # :206-208
    u = u.reshape(-1)[m]
    v = v.reshape(-1)[m]
    a, b, c, dd, e = inverse_intrinsics(KL)
    xn = ((a * u) + (b * v)) + c                                      # :219 row 0 of inv(K) @ [u;v;1]
    yn = (dd * v) + e                                                 # row 1
    x = xn * z                                                        # :221
    y = yn * z
    return x.astype(np.float32), y.astype(np.float32), z, mask


def warp_jacobian(x, y, z, KL) -> np.ndarray:
    """compute_jacobian_of_warp_function, utils/jacobian.py:7-44 (float32, operation order as
    written at :39-42; evaluated at the UN-warped point, quirk Q2).  Returns N x 2 x 6."""
    fx = F32(KL[0, 0])
    fy = F32(KL[1, 1])
    one = F32(1)
    n = x.shape[0]
    J = np.empty((n, 2, 6), dtype=np.float32)
    z2 = z * z
    J[:, 0, 0] = fx / z
    J[:, 0, 1] = 0.0
    J[:, 0, 2] = ((-fx) * x) / z2
    J[:, 0, 3] = ((-fx) * (x * y)) / z2
    J[:, 0, 4] = fx * (one + ((x * x) / z2))
    J[:, 0, 5] = ((-fx) * y) / z
    J[:, 1, 0] = 0.0
    J[:, 1, 1] = fy / z
    J[:, 1, 2] = ((-fy) * y) / z2
    J[:, 1, 3] = (-fy) * (one + ((y * y) / z2))
    J[:, 1, 4] = (fy * (x * y)) / z2
    J[:, 1, 5] = (fy * x) / z
    return J


def transform_points(T: np.ndarray, x, y, z):
    """np.dot(estimate.exp(), pointcloud), cpu_robust...:173 (float32 4x4 @ 4xN; w == 1)."""
    T = np.asarray(T, dtype=np.float32)
    xp = ((T[0, 0] * x + T[0, 1] * y) + T[0, 2] * z) + T[0, 3]
    yp = ((T[1, 0] * x + T[1, 1] * y) + T[1, 2] * z) + T[1, 3]
    zp = ((T[2, 0] * x + T[2, 1] * y) + T[2, 2] * z) + T[2, 3]
    return xp.astype(np.float32), yp.astype(np.float32), zp.astype(np.float32)


def project(KL: np.ndarray, xp, yp, zp):
    """RGBDCameraModel.project, camera_model.py:247-252: K_L(3x4) @ p' then divide by row 2."""
    fx, s, cx = F32(KL[0, 0]), F32(KL[0, 1]), F32(KL[0, 2])
    fy, cy = F32(KL[1, 1]), F32(KL[1, 2])
    with np.errstate(divide="ignore", invalid="ignore"):
        un = ((fx * xp) + (s * yp)) + (cx * zp)
        vn = (fy * yp) + (cy * zp)
        u = un / zp
        v = vn / zp
    return u.astype(np.float32), v.astype(np.float32)


def bilinear(image: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    """RobustDVOCPU.interpolate_bilinear, cpu_robust...:202-254, with the closed-boundary rule
    (SURVEY 8c): valid <=> 0<=x<=W-1 and 0<=y<=H-1, zero-weight taps never read, else NaN.
    Weights/sum in float64 (numba types int64 - float32 as float64), float32 store."""
    img = np.asarray(image)
    h, w = img.shape
    x = np.asarray(u, dtype=np.float32).astype(np.float64)
    y = np.asarray(v, dtype=np.float32).astype(np.float64)
    valid = (x >= 0.0) & (x <= w - 1) & (y >= 0.0) & (y <= h - 1)
    xs = np.where(valid, x, 0.0)
    ys = np.where(valid, y, 0.0)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    x1 = x0 + 1
    y1 = y0 + 1
    w00 = (x1 - xs) * (y1 - ys)
    w01 = (x1 - xs) * (ys - y0)
    w10 = (xs - x0) * (y1 - ys)
    w11 = (xs - x0) * (ys - y0)
    x1c = np.minimum(x1, w - 1)
    y1c = np.minimum(y1, h - 1)
    im = img.astype(np.float64)
    val = w00 * im[y0, x0] + w01 * im[y1c, x0] + w10 * im[y0, x1c] + w11 * im[y1c, x1c]
    return np.where(valid, val, np.nan).astype(np.float32)


# ================================================================================================
# Residuals + Jacobian (the backend hook)
# ================================================================================================
VARIANT_V1 = "v1"          # CPU default: grad I2 (Sobel) sampled at warped coords, J_w at un-warped p
VARIANT_V2 = "v2"          # CPU approximate_image2_gradient: grad I1 (Sobel) at the pixel
VARIANT_V3 = "v3"          # numba.cuda kernel: grad I1 central diff, J_w at WARPED p', strict bounds
VARIANT_GN = "gn"          # build-defined production mode (no reference counterpart): V1's residual / validity / sampled
                           # Sobel gradient of I2, normalised (x 1/8), J_w at the WARPED point p' (true Gauss-Newton row)


@dataclass
class PairLevel:
    """Inputs of one pyramid level of one frame pair."""
    gray_prev: np.ndarray
    depth_prev: np.ndarray
    gray_cur: np.ndarray
    KL: np.ndarray
    depth_scale: float
    gradx: Optional[np.ndarray] = None
    grady: Optional[np.ndarray] = None


def residuals_and_jacobian_v3(lv: PairLevel, T: np.ndarray):
    """The reference's numba.cuda backend: residuals_kernel (cuda/residuals_kernel.py:46-109) + the host-side
    compaction of gpu_robust_dense_visual_odometry.py:118-120, restated with the arithmetic types numba infers for
    that source: `tx - cx` is int32 - float32 = float64, so the back-projection, the rigid transform, the Jacobian
    terms and the projection are float64 (stored / passed on as float32 where the kernel's signature says float32:
    the Jacobian array, the arguments of the device function interpolate_bilinear); z = depth * depth_scale and the
    central-difference gradient (compute_gradients, :8-19) are float32.  No FMA contraction is assumed.
    PARITY UNPINNED: numba.cuda cannot run here and the reference holds no outputs of this kernel.

    Returns r (N,1) f32, J (N,6) f32, mask (H,W) bool — for this variant the mask the kernel writes is
    depth-valid AND strictly in bounds (x1 < W, y1 < H, :30-31), and equals `valid`."""
    d = np.asarray(lv.depth_prev)
    h, w = d.shape
    fx, fy = np.float32(lv.KL[0, 0]), np.float32(lv.KL[1, 1])
    cx, cy = np.float32(lv.KL[0, 2]), np.float32(lv.KL[1, 2])
    T = np.asarray(T, dtype=np.float32)
    R, tv = T[:3, :3], T[:3, 3]
    mask0 = d != 0                                                            # :60-65
    ty, tx = np.nonzero(mask0)
    z = d[mask0].astype(np.float32) * np.float32(lv.depth_scale)              # :67  uint16 * float32 -> float32
    z64 = z.astype(np.float64)
    x = (tx.astype(np.float64) - np.float64(cx)) * z64 / np.float64(fx)       # :69-70 float64
    y = (ty.astype(np.float64) - np.float64(cy)) * z64 / np.float64(fy)
    R64, t64 = R.astype(np.float64), tv.astype(np.float64)
    x1 = R64[0, 0] * x + R64[0, 1] * y + R64[0, 2] * z64 + t64[0]             # :73-75
    y1 = R64[1, 0] * x + R64[1, 1] * y + R64[1, 2] * z64 + t64[1]
    z1 = R64[2, 0] * x + R64[2, 1] * y + R64[2, 2] * z64 + t64[2]
    g1x, g1y = central_diff(lv.gray_prev)                                     # :82-83 (float32 return)
    gx = g1x[mask0].astype(np.float64)
    gy = g1y[mask0].astype(np.float64)
    fx64, fy64 = np.float64(fx), np.float64(fy)
    with np.errstate(divide="ignore", invalid="ignore"):
        J = np.empty((z.shape[0], 6), dtype=np.float32)                       # :87-92, float64 expressions, float32 store
        J[:, 0] = fx64 * gx / z1
        J[:, 1] = fy64 * gy / z1
        J[:, 2] = -fx64 * gx * x1 / (z1 ** 2) - fy64 * gy * y1 / (z1 ** 2)
        J[:, 3] = -fx64 * gx * x1 * y1 / (z1 ** 2) - fy64 * gy * ((y1 ** 2 / z1 ** 2) + 1)
        J[:, 4] = fx64 * gx * (x1 ** 2 / (z1 ** 2) + 1) + fy64 * gy * x1 * y1 / (z1 ** 2)
        J[:, 5] = -fx64 * gx * y1 / z1 + fy64 * gy * x1 / z1
        wx = (fx64 * x1 / z1 + np.float64(cx)).astype(np.float32)             # :95-96, then float32 argument of :99
        wy = (fy64 * y1 / z1 + np.float64(cy)).astype(np.float32)
        x0 = np.floor(wx)                                                     # x // 1
        y0 = np.floor(wy)
        inb = (x0 >= 0) & (y0 >= 0) & (x0 + 1 < w) & (y0 + 1 < h)             # :30-31 strict
    i2 = bilinear(lv.gray_cur, np.where(inb, wx, np.float32(-1)), np.where(inb, wy, np.float32(-1)))
    ok = ~np.isnan(i2)
    r = (i2[ok] - lv.gray_prev[mask0][ok].astype(np.float32)).reshape(-1, 1).astype(np.float32)   # :108
    valid = np.zeros((h, w), dtype=bool)
    valid[ty[ok], tx[ok]] = True
    return r, J[ok], valid, valid


def residuals_and_jacobian(lv: PairLevel, T: np.ndarray, variant: str = VARIANT_V1):
    """compute_residuals_and_jacobian, cpu_robust...:134-200 (V1/V2) or the numba.cuda kernel
    cuda/residuals_kernel.py:46-109 + host compaction gpu_robust...:118-120 (V3).

    Returns r (N,1) f32, J (N,6) f32, mask (H,W) bool [depth-valid only for V1/V2 — quirk Q6],
    valid (H,W) bool [depth-valid AND in-bounds: the pixels r/J rows correspond to]."""
    if variant == VARIANT_V3:
        return residuals_and_jacobian_v3(lv, T)
    x, y, z, mask = deproject(lv.depth_prev, lv.KL, lv.depth_scale)
    xp, yp, zp = transform_points(T, x, y, z)
    u, v = project(lv.KL, xp, yp, zp)
    h, w = lv.gray_cur.shape
    i2 = bilinear(lv.gray_cur, u, v)
    ok = ~np.isnan(i2)                                                   # :180
    i1 = lv.gray_prev[mask].astype(np.float32)
    r = (i2[ok] - i1[ok]).reshape(-1, 1).astype(np.float32)               # :191-194

    if variant == VARIANT_V1:
        if lv.gradx is None:
            lv.gradx, lv.grady = sobel3(lv.gray_cur)                     # _setup :58
        Jw = warp_jacobian(x[ok], y[ok], z[ok], lv.KL)                   # :168-170, :183
        gx = bilinear(lv.gradx, u[ok], v[ok])                            # :123-128
        gy = bilinear(lv.grady, u[ok], v[ok])
    elif variant == VARIANT_GN:
        # PARITY UNPINNED (build-defined, DVO_VARIANT_GN): the warp Jacobian of utils/jacobian.py:39-42 evaluated at the
        # warped point and multiplied by the unit-derivative gradient, in float64 (the device path is float32)
        if lv.gradx is None:
            lv.gradx, lv.grady = sobel3(lv.gray_cur)
        KL8 = np.array(lv.KL, dtype=np.float64)
        KL8[0, 0] *= 0.125
        KL8[1, 1] *= 0.125
        Jw = warp_jacobian(xp[ok].astype(np.float64), yp[ok].astype(np.float64), zp[ok].astype(np.float64), KL8)
        gx = bilinear(lv.gradx, u[ok], v[ok])
        gy = bilinear(lv.grady, u[ok], v[ok])
    elif variant == VARIANT_V2:
        g1x, g1y = sobel3(lv.gray_prev)                                  # :69-71
        Jw = warp_jacobian(x[ok], y[ok], z[ok], lv.KL)                   # :63-68 then row filter :188
        gx = g1x[mask][ok]
        gy = g1y[mask][ok]
    else:
        raise ValueError(variant)
    J = (gx[:, None] * Jw[:, 0, :] + gy[:, None] * Jw[:, 1, :]).astype(np.float32)   # _fill_jacobian :79-90
    valid = np.zeros(mask.shape, dtype=bool)
    flat = valid.reshape(-1)
    idx = np.flatnonzero(mask.reshape(-1))
    flat[idx[ok]] = True
    return r, J, mask, valid


# ================================================================================================
# Weights
# ================================================================================================
def student_t_scale_sum(r2: np.ndarray, lam: float, dof: int = 5) -> float:
    """TDistributionWeighter._compute_scale, weighter/t_weighter.py:36-47: a SUM (not a mean —
    quirk Q3; np.mean of a 1-element row), float64 accumulation as numba infers."""
    r2 = np.asarray(r2, dtype=np.float64).reshape(-1)
    return float(np.sum(r2 * ((dof + 1) / (dof + r2 * float(lam)))))


def student_t_weights(r: np.ndarray, dof: int = 5, initial_sigma: float = 5.0, tol: float = 1e-3,
                      max_iter: int = 50):
    """TDistributionWeighter.weight, weighter/t_weighter.py:21-34.  Returns (w float64, lambda)."""
    r2 = (np.asarray(r, dtype=np.float32) * np.asarray(r, dtype=np.float32)).astype(np.float64).reshape(-1)
    last = 1.0 / (initial_sigma ** 2)
    cur = last
    for _ in range(max_iter):
        sigma2 = student_t_scale_sum(r2, last, dof)
        cur = 1.0 / sigma2
        if abs(cur - last) < tol:
            break
        last = cur
    return (dof + 1) / (dof + r2 * cur), cur


def student_t_mean_weights(r: np.ndarray, lam_init: float, dof: int = 5, rel_tol: float = 1e-3, max_passes: int = 4):
    """Build-defined "proper" Student-t weighter (no reference counterpart beyond its intent; parity unpinned): the loop of
    weighter/t_weighter.py:21-34 with the scale taken as the MEAN of the weighted squared residuals (what :41-45 means to
    compute; as written `np.mean` runs over a 1-element row, i.e. it is a sum — quirk Q3), a relative stop rule
    |d lambda| <= rel_tol * lambda, at most `max_passes` passes, started from `lam_init` (the caller warm-starts it from
    the previous Gauss-Newton iteration).  Returns (w float64, lambda)."""
    r2 = (np.asarray(r, dtype=np.float32) * np.asarray(r, dtype=np.float32)).astype(np.float64).reshape(-1)
    n = r2.shape[0]
    last = float(lam_init)
    cur = last
    for _ in range(max_passes):
        sigma2 = student_t_scale_sum(r2, last, dof) / n
        cur = 1.0 / sigma2
        if abs(cur - last) <= rel_tol * cur:
            break
        last = cur
    return (dof + 1) / (dof + r2 * cur), cur


def huber_weights(r: np.ndarray, k: float):
    """Build-defined Huber weight w = min(1, k/|r|) (the reference has none; parity unpinned)."""
    a = np.abs(np.asarray(r, dtype=np.float64).reshape(-1))
    return np.where(a <= k, 1.0, k / np.maximum(a, 1e-300))


# ================================================================================================
# Normal equations + solve
# ================================================================================================
def normal_equations(r: np.ndarray, J: np.ndarray, w: Optional[np.ndarray] = None, accumulate: str = "fp64"):
    """base_robust_dvo.py:168-188.  H = J^T W J, b = -J^T W r, plus sum(w r^2) and N.
    accumulate="fp64": float64 accumulation of the float32 r/J (the value parity is checked
    against); "fp32": float32 BLAS exactly as the reference does."""
    n = r.shape[0]
    if accumulate == "fp32":
        Jt = np.ascontiguousarray(J.T)
        rr = r.astype(np.float32)
        JJ = J
        if w is not None:
            w32 = np.asarray(w, dtype=np.float32).reshape(-1, 1)
            err_sum = float(np.sum(w32 * (rr * rr), dtype=np.float32))
            rr = w32 * rr
            JJ = w32 * J
        else:
            err_sum = float(np.sum(rr * rr, dtype=np.float32))
        H = Jt @ JJ
        b = -(Jt @ rr)
        return H, b.reshape(6), err_sum, n
    J64 = J.astype(np.float64)
    r64 = r.astype(np.float64).reshape(-1)
    if w is not None:
        w64 = np.asarray(w, dtype=np.float64).reshape(-1)
        H = J64.T @ (w64[:, None] * J64)
        b = -(J64.T @ (w64 * r64))
        s = float(np.sum(w64 * r64 * r64))
    else:
        H = J64.T @ J64
        b = -(J64.T @ r64)
        s = float(np.sum(r64 * r64))
    return H, b, s, n


RANK_RCOND = float(np.finfo(np.float32).eps)


def solve_increment(H: np.ndarray, b: np.ndarray, rcond: float = RANK_RCOND) -> np.ndarray:
    """scipy.linalg.lstsq(H, b, lapack_driver="gelsy"), base_robust_dvo.py:196-198.  The reference calls it on a float32 H, so LAPACK's
    rank decision uses rcond = float32 eps: directions whose singular value is below 1.19e-7 of the largest carry no increment
    (minimum-norm solution).  Restated here on float64 data with that same rcond (pinned by tests/golden/rank_cases.npz)."""
    if not (np.all(np.isfinite(H)) and np.all(np.isfinite(b))):
        return np.full(6, np.nan)
    return scipy.linalg.lstsq(np.array(H, dtype=np.float64), np.array(b, dtype=np.float64).reshape(6, 1), cond=rcond, lapack_driver="gelsy",
                              check_finite=False)[0].reshape(6)


# ================================================================================================
# SE(3) / SO(3) (quaternion-backed, float32 storage) — utils/lie_algebra/*
# ================================================================================================
def wrap_angle(a):
    """common.py:32-46."""
    return (a + np.pi) % (2 * np.pi) - np.pi


def so3_quat_from_phi(phi: np.ndarray) -> np.ndarray:
    """So3.__init__ for a (3,1) vector, special_orthogonal_group.py:33-52 + _se3_to_quat :63-84."""
    phi = np.asarray(phi).reshape(3, 1)
    theta = np.linalg.norm(phi)
    if theta < LIE_EPSILON:
        return np.array([1.0, 0.0, 0.0, 0.0], dtype=np.float32)
    a = phi / theta
    theta = wrap_angle(theta)
    phi2 = theta * a
    th = np.linalg.norm(phi2)
    ax = phi2 / th
    s = math.sin(th / 2)
    xx, yy, zz = ax.flatten()
    return np.array([math.cos(th / 2), s * xx, s * yy, s * zz], dtype=np.float32)


def so3_matrix(q: np.ndarray) -> np.ndarray:
    """So3.exp, special_orthogonal_group.py:158-189 (float32 quaternion components)."""
    w, x, y, z = [F32(t) for t in np.asarray(q, dtype=np.float32).flatten()]
    return np.array([
        [2 * (w * w + x * x) - 1, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 2 * (w * w + y * y) - 1, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 2 * (w * w + z * z) - 1]], dtype=np.float32)


def so3_log(q: np.ndarray) -> np.ndarray:
    """So3.log, special_orthogonal_group.py:191-209 (note wrap_angle applied to theta/n)."""
    q = np.asarray(q, dtype=np.float32).flatten()
    w = q[0]
    vec = q[1:]
    n = np.linalg.norm(vec)
    if n < LIE_EPSILON:
        return np.zeros(3, dtype=np.float32)
    theta = wrap_angle(2 * math.atan2(n, w) / n)
    return (theta * vec)


def quat_from_matrix(R: np.ndarray) -> np.ndarray:
    """So3._SE3_to_quat, special_orthogonal_group.py:86-127 (float64 result)."""
    R = np.asarray(R)
    q = np.zeros(4)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(1 + t)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def quat_mult(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """common.py:51-73 (float32 result)."""
    a = np.asarray(a).reshape(4)
    b = np.asarray(b).reshape(4)
    w = a[0] * b[0] - np.dot(a[1:], b[1:])
    vec = a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])
    return np.concatenate(([w], vec)).astype(np.float32)


@dataclass
class Pose:
    """(quaternion wxyz float32 [or float64 after inverse()], translation) — mirrors Se3."""
    quat: np.ndarray
    tvec: np.ndarray

    @staticmethod
    def identity() -> "Pose":
        return Pose(np.array([1, 0, 0, 0], dtype=np.float32), np.zeros(3, dtype=np.float32))

    @staticmethod
    def from_xi(xi: np.ndarray) -> "Pose":
        """Se3.from_se3, special_euclidean_group.py:105-123."""
        xi = np.asarray(xi).reshape(6)
        ups = xi[:3].reshape(3, 1)
        q = so3_quat_from_phi(xi[3:])
        phi = so3_log(q)
        theta = np.linalg.norm(phi)
        if theta < LIE_EPSILON:
            return Pose(np.array([1, 0, 0, 0], dtype=np.float32), np.array(ups).reshape(3))
        ph = np.zeros((3, 3), dtype=np.float32)
        ph[0, 1], ph[0, 2], ph[1, 0], ph[1, 2], ph[2, 0], ph[2, 1] = -phi[2], phi[1], phi[2], -phi[0], -phi[1], phi[0]
        ph2 = np.dot(ph, ph)
        V = (np.eye(3, dtype=np.float32) + ((1 - math.cos(theta)) / (theta ** 2)) * ph +
             ((theta - math.sin(theta)) / (theta ** 3)) * ph2)
        return Pose(q, np.dot(V, ups).reshape(3))

    def matrix(self) -> np.ndarray:
        """Se3.exp, special_euclidean_group.py:35-52 -> 4x4 float32."""
        T = np.eye(4, dtype=np.float32)
        phi = so3_log(self.quat)
        if abs(np.linalg.norm(phi)) >= LIE_EPSILON:
            T[:3, :3] = so3_matrix(self.quat)
        T[:3, 3] = np.asarray(self.tvec).flatten()
        return T

    def mul(self, right: "Pose") -> "Pose":
        """Se3.__mul__, special_euclidean_group.py:91-96."""
        q = quat_mult(self.quat, right.quat)
        t = np.asarray(self.tvec).reshape(3) + np.dot(so3_matrix(self.quat), np.asarray(right.tvec).reshape(3))
        return Pose(q, t)

    def inverse(self) -> "Pose":
        """Se3.inverse :79-81 with So3.inverse via matrix->quaternion (:231-232)."""
        Rt = so3_matrix(self.quat).T.copy()
        q = quat_from_matrix(Rt)
        qf = q.astype(np.float32)
        Rinv = so3_matrix(qf)                       # So3.exp() evaluates float32-cast components
        return Pose(q, -np.dot(Rinv, np.asarray(self.tvec).reshape(3)))

    def log(self) -> np.ndarray:
        """Se3.log, special_euclidean_group.py:54-77 -> float32 (6,)."""
        xi = np.zeros(6, dtype=np.float32)
        phi = so3_log(self.quat)
        theta = np.linalg.norm(phi)
        if abs(theta) < LIE_EPSILON:
            xi[:3] = np.asarray(self.tvec).flatten()
            return xi
        a = (phi.reshape(3, 1) / theta)
        qa = so3_quat_from_phi(a)
        alog = so3_log(qa).reshape(3, 1)
        ah = np.zeros((3, 3), dtype=np.float32)
        al = alog.flatten()
        ah[0, 1], ah[0, 2], ah[1, 0], ah[1, 2], ah[2, 0], ah[2, 1] = -al[2], al[1], al[2], -al[0], -al[1], al[0]
        t2 = theta / 2
        A = t2 * np.cos(t2) / np.sin(t2)
        Vinv = A * np.eye(3, dtype=np.float32) + (1 - A) * np.dot(alog, alog.T) - t2 * ah
        xi[:3] = np.dot(Vinv, np.asarray(self.tvec).reshape(3, 1)).flatten()
        xi[3:] = phi.flatten()
        return xi


# ================================================================================================
# Gauss-Newton driver
# ================================================================================================
@dataclass
class IterRecord:
    level: int
    T: np.ndarray
    n: int
    H: np.ndarray
    b: np.ndarray
    inc: np.ndarray
    err: float
    sum_wr2: float


def gauss_newton(gray_prev, depth_prev, gray_cur, K, depth_scale, levels: int, max_iterations: int = 100,
                 tolerance: float = 1e-6, max_increased_steps_allowed: int = 0, use_weighter: bool = False,
                 variant: str = VARIANT_V1, accumulate: str = "fp64", weighting: Optional[str] = None,
                 huber_k: float = 1.345, init: Optional[Pose] = None, record: Optional[list] = None,
                 sigma: Optional[float] = None, last_transform: Optional[Pose] = None,
                 level_iterations: Optional[Sequence[int]] = None) -> Pose:
    """BaseRobustDVO._step, base_robust_dvo.py:137-236.  `level_iterations` (build-defined, dvo_set_level_iterations): iterations per
    level instead of one max_iterations for all.    `sigma` / `last_transform`: the motion prior of
    :69-71, :154, :190-194, :217-218 (`last_transform` = self._last_estimated_transform)."""
    gp = build_pyramid(gray_prev, levels)
    dp = build_pyramid(depth_prev, levels)
    gc = build_pyramid(gray_cur, levels)
    estimate = init if init is not None else Pose.identity()
    if weighting is None:
        weighting = "student_t_reference" if use_weighter else "none"
    sigma_prev = None
    for level in range(levels - 1, -1, -1):
        lam_t = 1.0 / 25.0                              # "student_t": every level starts from initial_sigma = 5 (t_weighter.py:11-15)
        lv = PairLevel(gp[level], dp[level], gc[level], level_intrinsics(K, level), depth_scale)
        err_prev = float(np.finfo("float32").max)
        increased = 0
        old = None
        if sigma is not None and last_transform is not None:
            old = Pose(np.array(last_transform.quat), np.array(last_transform.tvec))      # :154
        for i in range(max_iterations if level_iterations is None else int(level_iterations[min(level, len(level_iterations) - 1)])):
            T = estimate.matrix()
            r, J, _, _ = residuals_and_jacobian(lv, T, variant)
            n = r.shape[0]
            if weighting == "student_t_reference":
                w, _ = student_t_weights(r)
            elif weighting == "student_t":
                w, lam_t = student_t_mean_weights(r, lam_t)          # warm-started from the previous iteration of this level
            elif weighting == "huber":
                # build-defined (no reference counterpart): k = huber_k * sigma of the PREVIOUS evaluated
                # iteration of this pair (carried across levels); the very first iteration is unweighted
                w = None if sigma_prev is None else huber_weights(np.asarray(r, dtype=np.float32),
                                                                  float(np.float32(huber_k * sigma_prev)))
            else:
                w = None
            H, b, s, n = normal_equations(r, J, w, accumulate)
            if weighting == "huber":
                r64 = r.astype(np.float32).astype(np.float64).reshape(-1)
                sigma_prev = math.sqrt(float(np.sum(r64 * r64)) / n) if n else None
            err = float(np.float32(s / n)) if n else float("nan")             # :177 / :184 (float32 mean)
            if old is not None:                                               # :190-194
                xo = old.log().astype(np.float64)
                H = np.array(H, dtype=np.float64) + np.eye(6) / sigma
                b = np.array(b, dtype=np.float64).reshape(6) + xo / sigma
                err += 0.5 * sigma * float(np.linalg.norm(xo))
            inc_xi = solve_increment(H, b)
            if record is not None:
                record.append(IterRecord(level, T, n, np.array(H), np.array(b).reshape(6), np.array(inc_xi), err, s))
            inc = Pose.from_xi(inc_xi.astype(np.float32) if accumulate == "fp32" else inc_xi)
            diff = err - err_prev
            if abs(diff) < tolerance:                                         # :207
                break
            if diff < 0.0:                                                    # :212
                estimate = inc.mul(estimate)
                err_prev = err
                if old is not None:
                    old = inc.inverse().mul(old)                              # :217-218
                increased = 0
            else:
                increased += 1
            if increased > max_increased_steps_allowed:                       # :224
                break
    return estimate
