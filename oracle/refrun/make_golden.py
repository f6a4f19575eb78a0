#!/opt/conda/bin/python3.9
"""Golden-vector generator: executes the REFERENCE's own source files and records what they compute.

Run (build container only; /root/reference does not exist on the GPU box):

    PYTHONPATH=oracle/refrun/shims:/root/reference/src \
        /opt/conda/bin/python3.9 -W ignore oracle/refrun/make_golden.py

Environment: Python 3.9.7, NumPy 1.26.4, SciPy 1.7.1 (the only interpreter here whose NumPy
still has ``np.bool8``, which the reference needs at class-definition time).

What is the reference and what is not
-------------------------------------
Everything under ``dense_visual_odometry.*`` is imported from /root/reference/src unmodified:
the GN loop (core/robust_dense_visual_odometry/base_robust_dvo.py:137-236), the CPU backend
(cpu_robust_dense_visual_odometry.py:134-200), camera model, Lie classes, pyramid, weighter.
Not the reference (stand-ins, see oracle/refrun/shims): ``numba`` -> identity decorators,
``cv2`` -> NumPy restatements of cvtColor/medianBlur/Sobel (values parity unpinned except
Sobel, which the reference's own test pins).

One function is replaced: ``RobustDVOCPU.interpolate_bilinear``
(cpu_robust_dense_visual_odometry.py:202-254).  As written it sets NaN for out-of-bounds
pixels and then *falls through* to the 4-tap read (lines 241-252): undefined behaviour under
numba (unchecked indexing), IndexError under CPython.  It is replaced by the closed-boundary
rule: valid <=> 0 <= x <= W-1 and 0 <= y <= H-1; taps whose weight is zero are never read;
everything else is NaN.  Arithmetic follows numba's typing of the original source
(int64 - float32 -> float64 weights, float64 sum, float32 store).  This agrees with the
original wherever the original is well defined and passes the reference's own 10x10 test
(tests/.../test_cpu_robust_dense_visual_odometry.py:20-44: 100 valid pixels).

Three per-pixel Python loops are swapped for vectorised equivalents *after* being checked
against the reference loops on pyramid level 3 in `_check_vectorised_equivalents` (results in
meta.json): compute_jacobian_of_warp_function (utils/jacobian.py:7-44), _fill_jacobian
(cpu_robust...:79-90), TDistributionWeighter._compute_scale (weighter/t_weighter.py:36-47; the
vectorised form accumulates in float64 as numba's type inference does, the interpreted loop
accumulates in float32).

Outputs (tests/golden/): frames.npz (inputs), cli_compat.npz, pairs_l4.npz, single_l1.npz, weighted_l4.npz,
approx_l4.npz, sigma_l4.npz, dense_rj.npz, tiny10.npz, sequence.npz, lie.npz, bilinear_ref.npz, rank_cases.npz, meta.json.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import scipy
import scipy.linalg

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parents[2] / "tests" / "golden"

import cv2  # the stand-in  # noqa: E402
from dense_visual_odometry.core import get_dvo  # noqa: E402
from dense_visual_odometry.camera_model import RGBDCameraModel  # noqa: E402
from dense_visual_odometry.utils.lie_algebra import Se3, So3  # noqa: E402
from dense_visual_odometry.core.robust_dense_visual_odometry import base_robust_dvo as _brd  # noqa: E402
from dense_visual_odometry.core.robust_dense_visual_odometry import cpu_robust_dense_visual_odometry as _cpu  # noqa: E402
from dense_visual_odometry.core.robust_dense_visual_odometry.cpu_robust_dense_visual_odometry import RobustDVOCPU  # noqa: E402
from dense_visual_odometry.utils import jacobian as _jac  # noqa: E402
from dense_visual_odometry.weighter.t_weighter import TDistributionWeighter  # noqa: E402
from dense_visual_odometry.utils.image_pyramid import ImagePyramid  # noqa: E402


# ----------------------------------------------------------------------------------------------
# the one replaced function + the three vectorised equivalents
# ----------------------------------------------------------------------------------------------
def closed_boundary_bilinear(image, pixels_coordinates):
    img = np.asarray(image)
    h, w = img.shape
    xy = np.asarray(pixels_coordinates)
    assert xy.dtype == np.float32 and xy.ndim == 2 and xy.shape[1] == 2
    x = xy[:, 0].astype(np.float64)
    y = xy[:, 1].astype(np.float64)
    valid = (x >= 0.0) & (x <= w - 1) & (y >= 0.0) & (y <= h - 1)   # NaN coords compare False
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
    # taps with zero weight are not read: clamp their index (weight is exactly 0 there)
    x1c = np.minimum(x1, w - 1)
    y1c = np.minimum(y1, h - 1)
    im = img.astype(np.float64)
    val = (w00 * im[y0, x0] + w01 * im[y1c, x0] + w10 * im[y0, x1c] + w11 * im[y1c, x1c]) / 1.0
    out = np.where(valid, val, np.nan).astype(np.float32)
    return out.reshape(-1, 1)


def vec_warp_jacobian(pointcloud, calibration_matrix):
    fx = calibration_matrix[0, 0]
    fy = calibration_matrix[1, 1]
    x = pointcloud[0]
    y = pointcloud[1]
    z = pointcloud[2]
    n = pointcloud.shape[1]
    J = np.empty((n, 2, 6), dtype=np.float32)
    one = np.float32(1)
    J[:, 0, 0] = fx / z
    J[:, 0, 1] = 0.0
    J[:, 0, 2] = -fx * x / z ** 2
    J[:, 0, 3] = -fx * (x * y) / z ** 2
    J[:, 0, 4] = fx * (one + (x ** 2 / z ** 2))
    J[:, 0, 5] = -fx * y / z
    J[:, 1, 0] = 0.0
    J[:, 1, 1] = fy / z
    J[:, 1, 2] = -fy * y / z ** 2
    J[:, 1, 3] = -fy * (one + (y ** 2 / z ** 2))
    J[:, 1, 4] = fy * (x * y) / z ** 2
    J[:, 1, 5] = fy * x / z
    return J


def vec_fill_jacobian(Jw, gradients):
    g = np.asarray(gradients, dtype=np.float32)
    return (g[:, 0:1] * Jw[:, 0, :] + g[:, 1:2] * Jw[:, 1, :]).astype(np.float32)


def vec_compute_scale(residuals_squared, dof, last_lambda):
    r2 = np.asarray(residuals_squared, dtype=np.float64).reshape(-1)
    return float(np.sum(r2 * ((dof + 1) / (dof + r2 * float(last_lambda)))))


_ORIG = {}


def _check_vectorised_equivalents(frames, cam):
    """Run the reference's interpreted loops on level 3 of pair (1,2) and compare."""
    gray1, depth1 = frames[0]
    gray2, depth2 = frames[1]
    lvl = 3
    d = ImagePyramid(4, depth1).at(lvl)
    pc, mask = cam.deproject(d, return_mask=True, level=lvl)
    K = cam.at(lvl)
    ref = _ORIG["warp"](pointcloud=pc, calibration_matrix=K)
    vec = vec_warp_jacobian(pc, K)
    res = {"warp_jacobian_max_abs_diff": float(np.max(np.abs(ref - vec))),
           "warp_jacobian_bit_exact": bool(np.array_equal(ref, vec))}
    rng = np.random.default_rng(0)
    g = rng.integers(-1020, 1021, size=(ref.shape[0], 2)).astype(np.float32)
    g += rng.random(g.shape).astype(np.float32)
    a = _ORIG["fill"](ref, g)
    b = vec_fill_jacobian(ref, g)
    res["fill_jacobian_max_rel_diff"] = float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-30)))
    res["fill_jacobian_bit_exact"] = bool(np.array_equal(a, b))
    r2 = (rng.normal(0, 30, size=(ref.shape[0], 1)).astype(np.float32)) ** 2
    for lam in (1.0 / 25.0, 1e-7):
        s_ref = float(_ORIG["scale"](r2, 5, lam))
        s_vec = vec_compute_scale(r2, 5, lam)
        res["compute_scale_rel_diff_lambda_%g" % lam] = abs(s_ref - s_vec) / abs(s_vec)
    return res


def _check_bilinear_replacement(frames, cam, lvl=1, n_keep=6000):
    """The one replaced function against the reference's ORIGINAL `RobustDVOCPU.interpolate_bilinear`
    (cpu_robust_dense_visual_odometry.py:202-254, run here as plain Python: identity `njit`) on every warped
    coordinate of pyramid level `lvl` of pair (1,2) that is STRICTLY interior — x0 >= 0, y0 >= 0, x1 < W, y1 < H, i.e.
    where the original's bounds test (:241) does not fire and the function is well defined — for a uint8 image (the
    intensity lookup, :175-177) and a float32 plane (the Sobel gradient lookups, :123-128).  Must be bit-identical.
    The first `n_keep` coordinates with the ORIGINAL function's outputs are written to bilinear_ref.npz so the oracle's
    own `bilinear` is pinned to the reference function on the GPU box too."""
    gray1, depth1 = frames[0]
    gray2, _ = frames[1]
    d = ImagePyramid(4, depth1).at(lvl)
    img = ImagePyramid(4, gray2).at(lvl)
    gx = cv2.Sobel(img, cv2.CV_32F, 1, 0, ksize=3, borderType=cv2.BORDER_REFLECT)
    pc, _ = cam.deproject(d, return_mask=True, level=lvl)
    xi = np.array([0.0017, -0.0071, -0.0108, 0.0050, 0.0065, 0.0034], dtype=np.float32).reshape(6, 1)
    uv = cam.project(np.dot(Se3.from_se3(xi).exp(), pc), level=lvl)[:2, :].T.astype(np.float32)
    h, w = img.shape
    x0 = np.floor(uv[:, 0]).astype(np.int64)
    y0 = np.floor(uv[:, 1]).astype(np.int64)
    interior = (x0 >= 0) & (y0 >= 0) & (x0 + 1 < w) & (y0 + 1 < h)
    pts = np.ascontiguousarray(uv[interior])
    orig = _ORIG["bilinear"]
    res = {"level": lvl, "points_total": int(uv.shape[0]), "points_strictly_interior": int(pts.shape[0])}
    outs = {}
    for name, image in (("u8", img), ("f32", np.ascontiguousarray(gx, dtype=np.float32))):
        a = orig(image, pts)
        b = closed_boundary_bilinear(image, pts)
        assert a.dtype == np.float32 and a.shape == b.shape
        res[name + "_bit_exact"] = bool(np.array_equal(a, b)) and not bool(np.isnan(a).any())
        res[name + "_max_abs_diff"] = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))
        outs[name] = a
    np.savez_compressed(OUT / "bilinear_ref.npz", image_u8=img, image_f32=np.ascontiguousarray(gx, dtype=np.float32),
                        coords=pts[:n_keep], out_u8=outs["u8"][:n_keep], out_f32=outs["f32"][:n_keep])
    return res


def install_patches():
    _ORIG["warp"] = _jac.compute_jacobian_of_warp_function
    _ORIG["fill"] = RobustDVOCPU._fill_jacobian
    _ORIG["scale"] = TDistributionWeighter._compute_scale
    _ORIG["bilinear"] = RobustDVOCPU.interpolate_bilinear      # the reference's own function, kept for the check above
    RobustDVOCPU.interpolate_bilinear = staticmethod(closed_boundary_bilinear)


def install_fast_loops():
    _cpu.compute_jacobian_of_warp_function = vec_warp_jacobian
    _jac.compute_jacobian_of_warp_function = vec_warp_jacobian
    RobustDVOCPU._fill_jacobian = staticmethod(vec_fill_jacobian)
    TDistributionWeighter._compute_scale = staticmethod(vec_compute_scale)


# ----------------------------------------------------------------------------------------------
# capture machinery
# ----------------------------------------------------------------------------------------------
class Recorder:
    """Wraps the backend hook and the module-level `lstsq` used by BaseRobustDVO._step."""

    def __init__(self, dvo, keep_dense=False):
        self.dvo = dvo
        self.records = []
        self.keep_dense = keep_dense
        self.dense = []
        self._pending = None
        self._orig_hook = dvo.compute_residuals_and_jacobian
        self._orig_lstsq = _brd.lstsq
        dvo.compute_residuals_and_jacobian = self._hook
        _brd.lstsq = self._lstsq

    def close(self):
        _brd.lstsq = self._orig_lstsq
        self.dvo.compute_residuals_and_jacobian = self._orig_hook

    def _hook(self, estimate, level=0):
        r, J, mask = self._orig_hook(estimate=estimate, level=level)
        T = np.array(estimate.exp(), dtype=np.float32)
        self._pending = dict(level=level, T=T, r=r.copy(), J=J.copy(), mask_sum=int(mask.sum()))
        if self.keep_dense:
            self.dense.append(dict(level=level, T=T, r=r.copy(), J=J.copy(), mask=mask.copy()))
        return r, J, mask

    def _lstsq(self, a, b, **kw):
        H32 = np.array(a, dtype=np.float32)
        b32 = np.array(b, dtype=np.float32)
        out = self._orig_lstsq(a=a, b=b, **kw)
        p = self._pending
        r = p["r"].astype(np.float64)
        J = p["J"].astype(np.float64)
        w = None
        weighter = self.dvo._weighter
        if weighter is not None:
            w = weighter.weight(residuals_squared=(p["r"] * p["r"])).astype(np.float64)
            H64 = J.T @ (w * J)
            b64 = -(J.T @ (w * r))
            sw = float(np.sum(w * r * r))
        else:
            H64 = J.T @ J
            b64 = -(J.T @ r)
            sw = float(np.sum(r * r))
        inc64 = scipy.linalg.lstsq(H64, b64, lapack_driver="gelsy")[0]
        rec = dict(level=p["level"], T=p["T"], n=int(r.shape[0]), mask_sum=p["mask_sum"],
                   sum_wr2=sw, sum_r2=float(np.sum(r * r)),
                   H64=H64, b64=b64.reshape(6), inc64=inc64.reshape(6),
                   H32=H32, b32=b32.reshape(6), inc32=np.array(out[0], dtype=np.float32).reshape(6),
                   w_min=(float(w.min()) if w is not None else 1.0),
                   w_max=(float(w.max()) if w is not None else 1.0))
        self.records.append(rec)
        return out


def pack_records(records):
    keys = ["level", "n", "mask_sum", "sum_wr2", "sum_r2", "w_min", "w_max"]
    out = {k: np.array([r[k] for r in records]) for k in keys}
    for k in ["T", "H64", "b64", "inc64", "H32", "b32", "inc32"]:
        out[k] = np.stack([r[k] for r in records]) if records else np.zeros((0,))
    return out


def se3_to_arrays(se3):
    return dict(T=np.array(se3.exp(), dtype=np.float32), xi=np.array(se3.log(), dtype=np.float32).reshape(6),
                quat=np.array(se3.so3.quat, dtype=np.float32).reshape(4), tvec=np.array(se3.tvec, dtype=np.float32).reshape(3))


def run_pair(cam, prev, cur, keep_dense=False, **cfg):
    """One frame pair through the reference's public API: step(frame1) then step(frame2)."""
    dvo = get_dvo("robust-dvo", cam, Se3.identity(), **cfg)
    g1, d1 = prev
    g2, d2 = cur
    c1 = np.repeat(g1[..., None], 3, axis=2)    # equal channels -> BGR2GRAY returns g exactly
    c2 = np.repeat(g2[..., None], 3, axis=2)
    assert np.array_equal(cv2.cvtColor(c1, cv2.COLOR_BGR2GRAY), g1)
    dvo.step(color_image=c1, depth_image=d1.copy())
    rec = Recorder(dvo, keep_dense=keep_dense)
    try:
        tf = dvo.step(color_image=c2, depth_image=d2.copy())
    finally:
        rec.close()
    return tf, rec, dvo


def sigma_case(cam, frames, sigma=1e-13, n_frames=4):
    """Motion prior (base_robust_dvo.py:190-194, :217-218): chained step() over the first frames with `sigma` set.
    The prior couples consecutive pairs through `_last_estimated_transform`, so this is a sequential run."""
    dvo = get_dvo("robust-dvo", cam, Se3.identity(), levels=4, max_iterations=10, sigma=sigma)
    c = np.repeat(frames[0][0][..., None], 3, axis=2)
    dvo.step(color_image=c, depth_image=frames[0][1].copy())
    out = {"sigma": np.float64(sigma)}
    allrec, pair_idx, finals = [], [], []
    for k in range(1, n_frames):
        rec = Recorder(dvo)
        try:
            c = np.repeat(frames[k][0][..., None], 3, axis=2)
            tf = dvo.step(color_image=c, depth_image=frames[k][1].copy())
        finally:
            rec.close()
        allrec += rec.records
        pair_idx += [k - 1] * len(rec.records)
        finals.append(se3_to_arrays(tf))
    packed = pack_records(allrec)
    for key in ("level", "n", "T", "H32", "b32", "inc32", "H64", "b64"):      # H64/b64: data term only (no prior)
        out[key] = packed[key]
    out["pair"] = np.array(pair_idx)
    for key in ("T", "xi", "quat", "tvec"):
        out["final_" + key] = np.stack([f[key] for f in finals])
    np.savez_compressed(OUT / "sigma_l4.npz", **out)
    print("S sigma=%g: iterations per pair %s, xi=%s" % (sigma, np.bincount(out["pair"]).tolist(), out["final_xi"]), flush=True)


def cli_compat_case(cam, data_dir, gt):
    """The data flow of the reference's CLI on colour input (src/test_dvo.py:183 + core/base_dense_visual_odometry.py:58): the
    loader converts what imread returns to RGB and step() then applies BGR2GRAY to it, i.e. the red and blue weights are
    swapped.  Fixture frames 1, 2 through exactly those calls and the reference's step() (4 levels x 10 iterations):
    the gray planes the solver really sees and the transform it estimates -> cli_compat.npz."""
    rgb, depth = [], []
    for i in (1, 2):
        rgb.append(cv2.cvtColor(cv2.imread(str(data_dir / gt[str(i)]["rgb"]), cv2.IMREAD_ANYCOLOR), cv2.COLOR_BGR2RGB))     # test_dvo.py:183
        depth.append(cv2.imread(str(data_dir / gt[str(i)]["depth"]), cv2.IMREAD_UNCHANGED))                                  # :184
    dvo = get_dvo("robust-dvo", cam, Se3.identity(), levels=4, max_iterations=10)
    dvo.step(color_image=rgb[0], depth_image=depth[0].copy())
    tf = dvo.step(color_image=rgb[1], depth_image=depth[1].copy())
    a = se3_to_arrays(tf)
    np.savez_compressed(OUT / "cli_compat.npz", rgb_1=rgb[0], rgb_2=rgb[1], depth_1=depth[0], depth_2=depth[1],
                        gray_cli_1=cv2.cvtColor(rgb[0], cv2.COLOR_BGR2GRAY), gray_cli_2=cv2.cvtColor(rgb[1], cv2.COLOR_BGR2GRAY),
                        tf_T=a["T"], tf_xi=a["xi"])
    print("C cli-compat pair (1,2): xi=%s" % a["xi"], flush=True)


def rank_cases():
    """a15, the rank decision of the 6x6 solve (base_robust_dvo.py:196-198: scipy lstsq / gelsy on a float32 H, rcond = float32 eps).
    Three synthetic 160x120 pairs through the reference's own step(), single level, three iterations each:
      stripes   vertical stripes on a fronto-parallel plane: the image has no vertical gradient, column 1 of J is exactly zero, H has
                rank 5 under any rule;
      window    the same with a vertical gradient around two pixels only: cond(H) ~ 1e8, between the float32 rule (1.2e-7: rank 5) and a
                float64 rule (1e-13: rank 6) - the case in which the two rules give different increments;
      textured  stripes plus a weak horizontal texture everywhere: cond(H) ~ 1e5, full rank under both rules.
    Frame 2 is frame 1 shifted by one pixel in x.  Records H32 / b32 / inc32 (what the reference solved) and H64 / inc64 -> rank_cases.npz."""
    h, w = 120, 160
    yy, xx = np.mgrid[0:h, 0:w]
    K = np.array([[130.0, 0.0, 79.5], [0.0, 130.0, 59.5], [0.0, 0.0, 1.0]], dtype=np.float32)
    cam = RGBDCameraModel(K, 0.0002)
    depth = np.full((h, w), 10000, dtype=np.uint16)              # 2 m, fronto-parallel
    base = (127 + 90 * np.sin(xx * 0.35)).astype(np.float64)
    out = {"intrinsics": K, "depth_scale": np.float64(0.0002), "depth": depth}
    for name in ("stripes", "window", "textured"):
        g = base.copy()
        if name == "window":
            for (y, x) in [(41, 37), (83, 120)]:
                g[y, x] += 1.0                                     # two isolated one-grey-level bumps: the only vertical gradient of the image
        if name == "textured":
            g += 6.0 * np.sin(yy * 0.5)
        g1 = np.clip(np.rint(g), 0, 255).astype(np.uint8)
        g2 = np.roll(g1, 1, axis=1)
        tf, rec, _ = run_pair(cam, (g1, depth), (g2, depth), levels=1, max_iterations=3)
        a = pack_records(rec.records)
        for k in ("T", "n", "H32", "b32", "inc32", "H64", "b64", "inc64"):
            out["%s_%s" % (name, k)] = a[k]
        out[name + "_gray1"], out[name + "_gray2"] = g1, g2
        out[name + "_final_xi"] = se3_to_arrays(tf)["xi"]
        ev = np.linalg.eigvalsh(a["H64"][0])
        print("R %-8s iterations %d  eig(H64) min/max %.3g / %.3g = %.3g   inc32[0] %s\n%s inc64[0] %s" % (
            name, len(rec.records), ev[0], ev[-1], ev[0] / ev[-1], a["inc32"][0], " " * 44, a["inc64"][0]), flush=True)
    np.savez_compressed(OUT / "rank_cases.npz", **out)


def main():
    t_start = time.time()
    OUT.mkdir(parents=True, exist_ok=True)
    install_patches()
    if "--only-rank" in sys.argv:
        install_fast_loops()
        rank_cases()
        return
    if "--only-cli-compat" in sys.argv:
        install_fast_loops()
        data_dir = REF / "tests" / "test_data"
        with (data_dir / "ground_truth.json").open() as fp:
            gt = json.load(fp)
        cli_compat_case(RGBDCameraModel.load_from_yaml(data_dir / "camera_intrinsics.yaml"), data_dir, gt)
        return
    if "--only-sigma" in sys.argv:
        install_fast_loops()
        f = np.load(OUT / "frames.npz")
        cam = RGBDCameraModel(np.array(f["intrinsics"], dtype=np.float32), float(f["depth_scale"]))
        sigma_case(cam, [(f["gray"][i], f["depth"][i]) for i in range(10)])
        return

    data_dir = REF / "tests" / "test_data"
    with (data_dir / "ground_truth.json").open() as fp:
        gt = json.load(fp)
    cam = RGBDCameraModel.load_from_yaml(data_dir / "camera_intrinsics.yaml")

    frames = []
    colors = []
    gt_T = []
    for i in range(1, 11):
        bgr = cv2.imread(str(data_dir / gt[str(i)]["rgb"]), cv2.IMREAD_ANYCOLOR)
        depth = cv2.imread(str(data_dir / gt[str(i)]["depth"]), cv2.IMREAD_ANYDEPTH)
        gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
        frames.append((gray, depth))
        colors.append(bgr)
        gt_T.append(np.array(gt[str(i)]["transformation"], dtype=np.float64))

    meta = {
        "generator": "oracle/refrun/make_golden.py",
        "python": sys.version, "numpy": np.__version__, "scipy": scipy.__version__,
        "reference": "pfontana96/dense-visual-odometry @ /root/reference (VERSION %s)" % (REF / "VERSION").read_text().strip(),
        "stand_ins": ["numba -> identity decorators", "cv2 -> NumPy cvtColor/medianBlur/Sobel (parity unpinned except Sobel)"],
        "patched": ["RobustDVOCPU.interpolate_bilinear -> closed-boundary rule, float64 weights"],
        "gray_sums": [int(f[0].sum(dtype=np.int64)) for f in frames],
        "depth_sums": [int(f[1].sum(dtype=np.int64)) for f in frames],
    }

    meta["vectorised_loop_check"] = _check_vectorised_equivalents(frames, cam)
    print("vectorised loop check:", meta["vectorised_loop_check"], flush=True)
    meta["bilinear_replacement_check"] = _check_bilinear_replacement(frames, cam)
    print("bilinear replacement check:", meta["bilinear_replacement_check"], flush=True)
    assert meta["bilinear_replacement_check"]["u8_bit_exact"] and meta["bilinear_replacement_check"]["f32_bit_exact"]
    install_fast_loops()

    K = np.array(cam.intrinsics[:3, :3], dtype=np.float32)
    np.savez_compressed(
        OUT / "frames.npz",
        gray=np.stack([f[0] for f in frames]), depth=np.stack([f[1] for f in frames]),
        color_bgr_1=colors[0], intrinsics=K, depth_scale=np.float64(cam.depth_scale),
        gt_T=np.stack(gt_T),
        level_intrinsics=np.stack([cam.at(l) for l in range(5)]),
    )

    # ---- pyramids and gradients of frame 1/2 (inputs to the hot path; median/Sobel via stand-in) ----
    pyr = {}
    for name, img in (("gray1", frames[0][0]), ("depth1", frames[0][1]), ("gray2", frames[1][0])):
        p = ImagePyramid(4, img)
        for l in range(1, 4):
            pyr["%s_l%d" % (name, l)] = p.at(l)
    for l in range(4):
        g = frames[1][0] if l == 0 else pyr["gray2_l%d" % l]
        gx, gy = _jac.compute_gradients(image=g, kernel_size=3)
        pyr["gradx2_l%d" % l] = gx.astype(np.int16)
        pyr["grady2_l%d" % l] = gy.astype(np.int16)
        assert np.array_equal(gx, gx.astype(np.int16).astype(np.float32))
    np.savez_compressed(OUT / "pyramid_f1f2.npz", **pyr)

    # ---- A. config 1: pair (1,2), single level, run to convergence -------------------------------
    tf, rec, _ = run_pair(cam, frames[0], frames[1], levels=1)
    a = pack_records(rec.records)
    a.update({"final_" + k: v for k, v in se3_to_arrays(tf).items()})
    np.savez_compressed(OUT / "single_l1.npz", **a)
    print("A single level: %d iterations, xi=%s  (%.0fs)" % (len(rec.records), a["final_xi"], time.time() - t_start), flush=True)

    # ---- B. config 2 stand-in: 9 consecutive pairs, 4 levels x 10 iterations ----------------------
    allrec = []
    finals = []
    pair_idx = []
    for k in range(9):
        tf, rec, _ = run_pair(cam, frames[k], frames[k + 1], levels=4, max_iterations=10)
        allrec += rec.records
        pair_idx += [k] * len(rec.records)
        finals.append(se3_to_arrays(tf))
        print("B pair (%d,%d): %d iterations xi=%s (%.0fs)" % (k + 1, k + 2, len(rec.records), finals[-1]["xi"], time.time() - t_start), flush=True)
    b = pack_records(allrec)
    b["pair"] = np.array(pair_idx)
    for key in ("T", "xi", "quat", "tvec"):
        b["final_" + key] = np.stack([f[key] for f in finals])
    np.savez_compressed(OUT / "pairs_l4.npz", **b)

    # ---- C. Student-t weighter (reference quirk Q3), pair (1,2) ------------------------------------
    tf, rec, _ = run_pair(cam, frames[0], frames[1], levels=4, max_iterations=10, use_weighter=True)
    c = pack_records(rec.records)
    c.update({"final_" + k: v for k, v in se3_to_arrays(tf).items()})
    np.savez_compressed(OUT / "weighted_l4.npz", **c)
    print("C weighted: %d iterations xi=%s w in [%g,%g]" % (len(rec.records), c["final_xi"], c["w_min"].min(), c["w_max"].max()), flush=True)

    # ---- E. variant V2 (approximate_image2_gradient=True), pair (1,2) -----------------------------
    tf, rec, _ = run_pair(cam, frames[0], frames[1], levels=4, max_iterations=10, approximate_image2_gradient=True)
    e = pack_records(rec.records)
    e.update({"final_" + k: v for k, v in se3_to_arrays(tf).items()})
    np.savez_compressed(OUT / "approx_l4.npz", **e)
    print("E approx-gradient: %d iterations xi=%s" % (len(rec.records), e["final_xi"]), flush=True)

    # ---- F. dense r/J/mask at levels 3 and 2 (identity and a non-identity estimate) ---------------
    dvo = get_dvo("robust-dvo", cam, Se3.identity(), levels=4)
    c1 = np.repeat(frames[0][0][..., None], 3, axis=2)
    dvo.step(color_image=c1, depth_image=frames[0][1].copy())
    dvo._build_pyramids(gray_image=frames[1][0], depth_image=frames[1][1])
    dense = {}
    xi_test = np.array([0.0017, -0.0071, -0.0108, 0.0050, 0.0065, 0.0034], dtype=np.float32).reshape(6, 1)
    for lvl in (3, 2):
        dvo._setup(level=lvl)
        for tag, est in (("id", Se3.identity()), ("xi", Se3.from_se3(xi_test))):
            r, J, mask = dvo.compute_residuals_and_jacobian(estimate=est, level=lvl)
            pre = "l%d_%s_" % (lvl, tag)
            dense[pre + "T"] = np.array(est.exp(), dtype=np.float32)
            dense[pre + "r"] = r
            dense[pre + "J"] = J
            dense[pre + "mask"] = mask
        dvo._cleanup()
    np.savez_compressed(OUT / "dense_rj.npz", **dense)

    # ---- D. the reference's own 10x10 known-answer case (test_cpu_robust...:20-44) -----------------
    h, w = 10, 10
    gray = np.full((h, w), 150, dtype=np.uint8)
    gray[: h // 2, : w // 2] = 150 / 3.0
    color = np.repeat(gray[..., None], 3, axis=2)
    depth = np.ones_like(gray)
    depth[: h // 2, : w // 2] = 3.0
    cam10 = RGBDCameraModel(np.eye(3, dtype=np.float32), 1.0)
    dvo = get_dvo("robust-dvo", cam10, Se3.identity(), levels=1)
    dvo.step(color_image=color, depth_image=depth)
    dvo._build_pyramids(gray_image=gray, depth_image=depth)
    dvo._setup(level=0)
    r, J, mask = dvo.compute_residuals_and_jacobian(estimate=Se3.identity(), level=0)
    assert r.shape == (100, 1) and np.all(r == 0)
    np.savez_compressed(OUT / "tiny10.npz", gray=gray, depth=depth, r=r, J=J, mask=mask)

    # ---- G. chained step() over the ten frames (API-level trajectory) -------------------------------
    init = Se3(So3(gt_T[0][:3, :3]), gt_T[0][:3, 3].reshape(3, 1))
    dvo = get_dvo("robust-dvo", cam, init, levels=4, max_iterations=10)
    poses = []
    tfs = []
    for k in range(10):
        c = np.repeat(frames[k][0][..., None], 3, axis=2)
        tf = dvo.step(color_image=c, depth_image=frames[k][1].copy())
        tfs.append(se3_to_arrays(tf))
        poses.append(se3_to_arrays(dvo.current_pose))
    np.savez_compressed(OUT / "sequence.npz",
                        pose_T=np.stack([p["T"] for p in poses]), pose_xi=np.stack([p["xi"] for p in poses]),
                        tf_T=np.stack([p["T"] for p in tfs]), tf_xi=np.stack([p["xi"] for p in tfs]),
                        init_T=np.array(init.exp(), dtype=np.float32))
    print("G sequence done (%.0fs)" % (time.time() - t_start), flush=True)

    # ---- H. Lie-group vectors (pure-NumPy reference classes, imported unmodified) ------------------
    rng = np.random.default_rng(7)
    xis = np.concatenate([rng.normal(0, 0.3, size=(24, 6)), rng.normal(0, 1e-3, size=(8, 6)), np.zeros((1, 6)),
                          np.array([[0.1, -0.2, 0.3, 0, 0, 0]])]).astype(np.float32)
    lie = dict(xi=xis, T=[], log=[], inv_T=[], quat=[], comp_T=[])
    prev = Se3.identity()
    for x in xis:
        s = Se3.from_se3(x.reshape(6, 1))
        lie["T"].append(np.array(s.exp(), dtype=np.float32))
        lie["log"].append(np.array(s.log(), dtype=np.float32).reshape(6))
        lie["inv_T"].append(np.array(s.inverse().exp(), dtype=np.float32))
        lie["quat"].append(np.array(s.so3.quat, dtype=np.float32).reshape(4))
        lie["comp_T"].append(np.array((s * prev).exp(), dtype=np.float32))
        prev = s
    np.savez_compressed(OUT / "lie.npz", **{k: np.stack(v) if isinstance(v, list) else v for k, v in lie.items()})

    sigma_case(cam, frames)
    cli_compat_case(cam, data_dir, gt)
    rank_cases()

    meta["elapsed_s"] = time.time() - t_start
    with (OUT / "meta.json").open("w") as fp:
        json.dump(meta, fp, indent=1)
    print("done in %.0fs" % meta["elapsed_s"])


if __name__ == "__main__":
    main()
