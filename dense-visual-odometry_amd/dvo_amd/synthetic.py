"""Seeded synthetic RGB-D streams of the shape BASELINE.json's configs name (no datasets travel to the
GPU box, and TUM fr1/desk / fr2/desk are not on disk): a textured, tilted plane with smooth relief seen by
a pinhole camera moving along a smooth trajectory.  Every frame is rendered exactly (ray / surface
intersection by fixed-point iteration), so consecutive frames are photo-consistent and the known camera
motion is the ground truth.

Statistics follow SURVEY 8d config 3: intensity = several octaves of smooth texture scaled to u8, depth in
[0.8 m, 4.5 m] quantised to u16 with depth_scale 0.0002, a 16-px zero border and ~20 % zero blobs
(overall depth-valid fraction ~0.75, like the TUM fixtures).
"""
from __future__ import annotations

import numpy as np

TUM_FR1_K = np.array([[517.3, 0.0, 318.6], [0.0, 516.5, 239.5], [0.0, 0.0, 1.0]], dtype=np.float32)
DEPTH_SCALE = 0.0002


def scaled_intrinsics(width: int, height: int) -> np.ndarray:
    """TUM fr1 intrinsics scaled from 640x480 to width x height."""
    sx, sy = width / 640.0, height / 480.0
    K = TUM_FR1_K.astype(np.float64).copy()
    K[0, 0] *= sx
    K[0, 2] = (K[0, 2] + 0.5) * sx - 0.5
    K[1, 1] *= sy
    K[1, 2] = (K[1, 2] + 0.5) * sy - 0.5
    return K.astype(np.float32)


def _texture(X, Y, params):
    import torch
    val = torch.zeros_like(X)
    for (fx, fy, ph1, ph2, amp) in params:
        val += amp * torch.sin(X * fx + ph1) * torch.cos(Y * fy + ph2)
    return val


def _relief(X, Y):
    import torch
    return 0.25 * torch.sin(X * 1.3) * torch.cos(Y * 1.1) + 0.1 * torch.sin(X * 3.1 + 0.5)


def _rot(omega):
    th = np.linalg.norm(omega)
    if th < 1e-12:
        return np.eye(3)
    k = omega / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def make_stream(n_frames: int, width: int = 640, height: int = 480, seed: int = 0, step_trans: float = 0.01,
                step_rot_deg: float = 0.5, invalid_fraction: float = 0.2, device=None, as_torch: bool = False):
    """-> dict(gray u8 [n,h,w], depth u16 [n,h,w], K f32 3x3, depth_scale, T_wc f64 [n,4,4] camera-to-world).

    Rendered with torch in float64 on `device` (CPU by default; pass "cuda" on the GPU box to build long
    streams quickly).  `as_torch=True` leaves gray (uint8) / depth (int16 bit pattern of the u16 values) as
    device tensors; otherwise NumPy arrays are returned."""
    import torch
    dev = torch.device(device) if device is not None else torch.device("cpu")
    rng = np.random.default_rng(seed)
    K = scaled_intrinsics(width, height).astype(np.float64)
    octaves = []
    for o in range(6):
        f = 2.2 * (1.9 ** o)
        octaves.append((f * rng.uniform(0.8, 1.2), f * rng.uniform(0.8, 1.2), rng.uniform(0, 6.28),
                        rng.uniform(0, 6.28), 1.0 / (1.35 ** o)))
    norm = sum(a[4] for a in octaves)
    v, u = torch.meshgrid(torch.arange(height, dtype=torch.float64, device=dev),
                          torch.arange(width, dtype=torch.float64, device=dev), indexing="ij")
    rays = torch.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)], dim=0).reshape(3, -1)

    # bounded smooth trajectory (the scene stays in view and in depth range for any stream length): position and
    # rotation vector are sums of sinusoids whose per-frame steps are ~step_trans metres and ~step_rot_deg degrees
    amp_t = np.array([0.25, 0.15, 0.10])
    amp_r = np.array([0.06, 0.06, 0.04])
    per_t = 2 * np.pi * amp_t / (step_trans / np.sqrt(3)) * rng.uniform(0.9, 1.1, 3)
    per_r = 2 * np.pi * amp_r / (np.deg2rad(step_rot_deg) / np.sqrt(3)) * rng.uniform(0.9, 1.1, 3)
    ph_t = rng.uniform(0, 2 * np.pi, 3)
    ph_r = rng.uniform(0, 2 * np.pi, 3)

    def pose_at(k):
        tk = amp_t * (np.sin(2 * np.pi * k / per_t + ph_t) - np.sin(ph_t))
        rk = amp_r * (np.sin(2 * np.pi * k / per_r + ph_r) - np.sin(ph_r))
        return _rot(rk), tk

    grays, depths, poses = [], [], []
    z0 = 2.4
    b = max(1, int(round(16 * width / 640.0)))
    blob_params = [(1.0, 1.0, 0.3, 1.1, 1.0), (2.3, 1.7, 2.0, 0.4, 0.6)]
    for k in range(n_frames):
        R, t = pose_at(k)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        poses.append(T)
        Rt = torch.as_tensor(R, dtype=torch.float64, device=dev)
        tt = torch.as_tensor(t, dtype=torch.float64, device=dev)
        dw = Rt @ rays                                 # world ray directions (camera z-forward)
        # surface: Z = z0 + 0.35*X + 0.2*Y + relief(X, Y); solve s: (t + s*dw).z = surface((t+s*dw).xy)
        s = torch.full((rays.shape[1],), z0, dtype=torch.float64, device=dev)
        for _ in range(12):
            P = tt[:, None] + s * dw
            target = z0 + 0.35 * P[0] + 0.2 * P[1] + _relief(P[0], P[1])
            s = s + (target - P[2]) / torch.clamp(dw[2], min=0.2)
        P = tt[:, None] + s * dw
        tex = _texture(P[0], P[1], octaves) / norm
        gray = torch.clamp(127.5 + 127.5 * 1.6 * tex, 0, 255).round().to(torch.uint8).reshape(height, width)
        zc = s.reshape(height, width)                  # camera-frame depth (ray z component is 1)
        depth = (torch.clamp(zc, 0.8, 4.5) / DEPTH_SCALE).round().to(torch.int32)
        blob = _texture(u * 0.045 + 13.0 * k, v * 0.05, blob_params)
        thr = torch.quantile(blob.reshape(-1)[:: max(1, blob.numel() // 200000)], 1.0 - invalid_fraction)
        depth[blob > thr] = 0
        depth[:b], depth[-b:], depth[:, :b], depth[:, -b:] = 0, 0, 0, 0
        grays.append(gray)
        depths.append(depth.to(torch.int16))            # bit pattern of the u16 value (max 22500 < 32768)
    gray_t = torch.stack(grays).contiguous()
    depth_t = torch.stack(depths).contiguous()
    out = dict(K=K.astype(np.float32), depth_scale=DEPTH_SCALE, T_wc=np.stack(poses))
    if as_torch:
        out["gray"], out["depth"] = gray_t, depth_t
    else:
        out["gray"] = gray_t.cpu().numpy()
        out["depth"] = depth_t.cpu().numpy().view(np.uint16)
    return out


def relative_transform(T_wc_prev: np.ndarray, T_wc_cur: np.ndarray) -> np.ndarray:
    """Ground-truth transform of points from the previous camera frame into the current one
    ({t-1}_to_{t} in the reference's words, base_dense_visual_odometry.py:75-79)."""
    return np.linalg.inv(T_wc_cur) @ T_wc_prev
