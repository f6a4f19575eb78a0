"""TUM RGB-D benchmark I/O around the hot path: sequence loader with RGB / depth / ground-truth
timestamp association, TUM-format trajectory writer, ATE / RPE evaluation.

Restates the data handling of the reference's CLI harness (src/test_dvo.py:86-206 loader, :337-345 trajectory
writer) without its defects: images are decoded to BGR and go through the BGR2GRAY ingest as BGR (the reference
feeds RGB to a BGR conversion, src/test_dvo.py:183 + core/base_dense_visual_odometry.py:58), and frames are
streamed instead of being held in RAM all at once.  The reference has no ATE/RPE evaluator; the one here follows
the TUM benchmark's definition (rigid alignment of positions, RMSE).
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Iterator, List, Optional, Tuple

import numpy as np

from .batch import absolute_trajectory_error


@dataclass
class TumSequence:
    root: Path
    timestamps: np.ndarray        # (n,) mean of the associated rgb / depth stamps
    rgb_paths: List[str]
    depth_paths: List[str]
    gt_T_wc: Optional[np.ndarray]  # (n,4,4) camera-to-world ground truth nearest in time, or None
    gt_timestamps: Optional[np.ndarray]

    def __len__(self):
        return len(self.rgb_paths)


def _read_list(path: Path):
    stamps, cols = [], []
    with path.open("r") as fp:
        for line in fp:
            if line.startswith("#") or not line.strip():
                continue
            parts = line.rstrip("\r\n").split()
            stamps.append(float(parts[0]))
            cols.append(parts[1:])
    return np.asarray(stamps, dtype=np.float64), cols


def quat_xyzw_to_matrix(q) -> np.ndarray:
    x, y, z, w = (float(t) for t in q)
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n if n > 0 else 0.0
    return np.array([[1 - s * (y * y + z * z), s * (x * y - w * z), s * (x * z + w * y)],
                     [s * (x * y + w * z), 1 - s * (x * x + z * z), s * (y * z - w * x)],
                     [s * (x * z - w * y), s * (y * z + w * x), 1 - s * (x * x + y * y)]])


def matrix_to_quat_xyzw(R) -> np.ndarray:
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0, 0.0, 0.0, (R[k, j] - R[j, k]) / s]
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    return np.asarray(q, dtype=np.float64)


def load_tum_sequence(root: Path, size: Optional[int] = None) -> TumSequence:
    """rgb.txt / depth.txt / groundtruth.txt association as in src/test_dvo.py:113-170: every RGB frame takes
    its closest depth frame, duplicates (two RGB frames claiming one depth frame) keep the first, and the ground
    truth pose closest to the mean of the two stamps is attached."""
    root = Path(root)
    for name in ("rgb.txt", "depth.txt"):
        if not (root / name).exists():
            raise FileNotFoundError("Expected TUM RGB-D dataset to contain a file named '{}' at '{}'".format(name, root))
    t_rgb, c_rgb = _read_list(root / "rgb.txt")
    t_d, c_d = _read_list(root / "depth.txt")
    closest = np.abs(t_rgb.reshape(-1, 1) - t_d.reshape(1, -1)).argmin(axis=1)
    depth_idx, rgb_idx = np.unique(closest, return_index=True)        # first RGB frame per depth frame
    rgb_paths = [str(root / c_rgb[i][0]) for i in rgb_idx]
    depth_paths = [str(root / c_d[j][0]) for j in depth_idx]
    stamps = (t_rgb[rgb_idx] + t_d[depth_idx]) / 2
    gt_T = gt_t = None
    if (root / "groundtruth.txt").exists():
        t_gt, c_gt = _read_list(root / "groundtruth.txt")
        near = np.abs(stamps.reshape(-1, 1) - t_gt.reshape(1, -1)).argmin(axis=1)
        gt_T = np.tile(np.eye(4), (len(stamps), 1, 1))
        for n, g in enumerate(near):
            vals = [float(v) for v in c_gt[g]]
            gt_T[n, :3, :3] = quat_xyzw_to_matrix(vals[3:7])
            gt_T[n, :3, 3] = vals[0:3]
        gt_t = t_gt[near]
    if size is not None and size < len(rgb_paths):
        rgb_paths, depth_paths, stamps = rgb_paths[:size], depth_paths[:size], stamps[:size]
        if gt_T is not None:
            gt_T, gt_t = gt_T[:size], gt_t[:size]
    return TumSequence(root, stamps, rgb_paths, depth_paths, gt_T, gt_t)


def read_frame(rgb_path: str, depth_path: str, channel_order: str = "bgr") -> Tuple[np.ndarray, np.ndarray]:
    """-> (colour u8 HxWx3, depth u16 HxW).  channel_order "bgr" (default): the array layout cv2.imread gives, which is what
    the BGR2GRAY ingest expects.  "rgb": what the reference's CLI really hands to step() — its loader converts the image to RGB
    (src/test_dvo.py:183) and step() then applies BGR2GRAY to it (core/base_dense_visual_odometry.py:58), i.e. red and blue
    weights swapped; only for like-for-like comparisons with runs of that CLI (tools/run_sequence.py --reference-cli-compat)."""
    if channel_order not in ("bgr", "rgb"):
        raise ValueError("channel_order must be 'bgr' or 'rgb'")
    from PIL import Image
    rgb = np.array(Image.open(rgb_path).convert("RGB"))
    depth = np.array(Image.open(depth_path)).astype(np.uint16)
    return (np.ascontiguousarray(rgb) if channel_order == "rgb" else np.ascontiguousarray(rgb[..., ::-1])), depth


def write_tum_dataset(root: Path, frames, depths, T_wc=None, t0: float = 1305031102.0, fps: float = 30.0, depth_lag: float = 0.004,
                      n_listed: Optional[int] = None, K=None, depth_scale: Optional[float] = None) -> Path:
    """Writes a dataset in the TUM RGB-D layout (rgb/*.png 8-bit colour, depth/*.png 16-bit, rgb.txt / depth.txt / groundtruth.txt with
    unsynchronised stamps as a Kinect gives them, optional camera_intrinsics.yaml): what `load_tum_sequence` and the reference's
    loader (src/test_dvo.py:86-206) read.  `frames`: (n,h,w) gray or (n,h,w,3) BGR u8; `depths` (n,h,w) u16.  `n_listed` > n lists the
    n files again and again, forwards then backwards (consecutive entries stay consecutive renders), under new stamps: a long
    sequence for decode-throughput measurements without encoding every frame."""
    from PIL import Image
    root = Path(root)
    (root / "rgb").mkdir(parents=True, exist_ok=True)
    (root / "depth").mkdir(exist_ok=True)
    frames, depths = np.asarray(frames), np.asarray(depths)
    n = frames.shape[0]
    names = []
    for k in range(n):
        tr, td = t0 + k / fps, t0 + k / fps + depth_lag
        img = frames[k] if frames.ndim == 4 else np.repeat(frames[k][..., None], 3, axis=2)
        rn, dn = "rgb/%.6f.png" % tr, "depth/%.6f.png" % td
        Image.fromarray(np.ascontiguousarray(img[..., ::-1]) if frames.ndim == 4 else img).save(root / rn)       # file holds RGB
        Image.fromarray(depths[k].astype(np.uint16)).save(root / dn)
        names.append((rn, dn))
    total = n if n_listed is None else int(n_listed)
    period = max(2 * (n - 1), 1)
    with (root / "rgb.txt").open("w") as fr, (root / "depth.txt").open("w") as fd:
        fr.write("# color images\n# file: 'synthetic'\n# timestamp filename\n")
        fd.write("# depth maps\n# timestamp filename\n")
        gt = (root / "groundtruth.txt").open("w") if T_wc is not None else None
        if gt:
            gt.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
        for i in range(total):
            k = i % period
            k = k if k < n else period - k
            tr, td = t0 + i / fps, t0 + i / fps + depth_lag
            fr.write("%.6f %s\n" % (tr, names[k][0]))
            fd.write("%.6f %s\n" % (td, names[k][1]))
            if gt:
                T = np.asarray(T_wc[k])
                q = matrix_to_quat_xyzw(T[:3, :3])
                gt.write("%.6f %r %r %r %r %r %r %r\n" % ((tr + depth_lag / 2,) + tuple(float(v) for v in T[:3, 3]) + tuple(float(v) for v in q)))
        if gt:
            gt.close()
    if K is not None and depth_scale is not None:
        K = np.asarray(K, dtype=np.float64)
        (root / "camera_intrinsics.yaml").write_text("intrinsics:\n" + "".join("  - [%r, %r, %r]\n" % tuple(float(v) for v in row) for row in K[:3, :3]) +
                                                     "depth_scale: %r\n" % float(depth_scale))
    return root


def iter_frames(seq: TumSequence, channel_order: str = "bgr") -> Iterator[Tuple[np.ndarray, np.ndarray]]:
    for rp, dp in zip(seq.rgb_paths, seq.depth_paths):
        yield read_frame(rp, dp, channel_order)


def write_tum_trajectory(path: Path, timestamps, T_wc) -> None:
    """`timestamp tx ty tz qx qy qz qw` per pose (src/test_dvo.py:337-345)."""
    with Path(path).open("w") as fp:
        fp.write("# timestamp tx ty tz qx qy qz qw\n")
        for ts, T in zip(timestamps, T_wc):
            q = matrix_to_quat_xyzw(np.asarray(T)[:3, :3])
            t = np.asarray(T)[:3, 3]
            fp.write(" ".join(repr(float(v)) for v in [ts, *t, *q]) + "\n")


def read_tum_trajectory(path: Path):
    stamps, cols = _read_list(Path(path))
    T = np.tile(np.eye(4), (len(stamps), 1, 1))
    for n, c in enumerate(cols):
        vals = [float(v) for v in c]
        T[n, :3, :3] = quat_xyzw_to_matrix(vals[3:7])
        T[n, :3, 3] = vals[0:3]
    return stamps, T


def relative_pose_error(est_T_wc: np.ndarray, gt_T_wc: np.ndarray, delta: int = 1) -> Tuple[float, float]:
    """TUM RPE over frame distance `delta`: RMSE of the translational (m) and rotational (rad) parts of
    (Q_i^-1 Q_{i+d})^-1 (P_i^-1 P_{i+d})."""
    et, er = [], []
    for i in range(len(est_T_wc) - delta):
        dp = np.linalg.inv(est_T_wc[i]) @ est_T_wc[i + delta]
        dq = np.linalg.inv(gt_T_wc[i]) @ gt_T_wc[i + delta]
        e = np.linalg.inv(dq) @ dp
        et.append(np.linalg.norm(e[:3, 3]))
        er.append(np.arccos(np.clip((np.trace(e[:3, :3]) - 1) / 2, -1.0, 1.0)))
    if not et:
        return float("nan"), float("nan")
    return float(np.sqrt(np.mean(np.square(et)))), float(np.sqrt(np.mean(np.square(er))))


def evaluate(est_T_wc: np.ndarray, gt_T_wc: np.ndarray) -> dict:
    rt, rr = relative_pose_error(est_T_wc, gt_T_wc, 1)
    return {"ate_rmse_m": absolute_trajectory_error(est_T_wc, gt_T_wc), "rpe_trans_rmse_m": rt, "rpe_rot_rmse_rad": rr,
            "frames": int(len(est_T_wc))}
