"""ctypes binding of libdvo_hip.so — the C ABI declared in include/dvo_hip.h.

There is no fallback: if the shared library is missing the import of this module raises, and if no
HIP device is present `HipContext(...)` raises `DVOHipError`.  Build the library with
`python __graft_entry__.py` (or `make -C dense-visual-odometry_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

NE_STRIDE = 32
TRACE_STRIDE = 50
MAX_LEVELS = 8

VARIANT_V1, VARIANT_V2, VARIANT_V3, VARIANT_GN = 0, 1, 2, 3
WEIGHT_NONE, WEIGHT_STUDENT_T, WEIGHT_HUBER, WEIGHT_STUDENT_T_MEAN = 0, 1, 2, 3

import os as _os

# DVO_HIP_LIB: alternative build of the same library (kernel-tuning experiments); default = in-tree build
_LIB_PATH = Path(_os.environ.get("DVO_HIP_LIB") or (Path(__file__).resolve().parent / "libdvo_hip.so"))

# every symbol include/dvo_hip.h declares: name -> (restype, argtypes)
_c_int_p = C.POINTER(C.c_int)
_c_float_p = C.POINTER(C.c_float)
_c_double_p = C.POINTER(C.c_double)
_c_u8_p = C.POINTER(C.c_uint8)
_c_u16_p = C.POINTER(C.c_uint16)
_c_i16_p = C.POINTER(C.c_int16)
_c_i64_p = C.POINTER(C.c_int64)
_ctx = C.c_void_p

SIGNATURES = {
    "dvo_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_ctx)]),
    "dvo_destroy": (None, [_ctx]),
    "dvo_last_error": (C.c_char_p, [_ctx]),
    "dvo_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "dvo_synchronize": (C.c_int, [_ctx]),
    "dvo_set_camera": (C.c_int, [_ctx, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_double]),
    "dvo_get_level_intrinsics": (C.c_int, [_ctx, C.c_int, _c_float_p]),
    "dvo_upload_frame": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dvo_upload_frame_device": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dvo_upload_frames_device": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dvo_upload_frames": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dvo_wait_for": (C.c_int, [_ctx, _ctx]),
    "dvo_set_upload_peer": (C.c_int, [_ctx, _ctx]),
    "dvo_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "dvo_host_free": (C.c_int, [C.c_void_p]),
    "dvo_upload_frame_bgr": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]),
    "dvo_upload_frames_bgr": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]),
    "dvo_upload_frames_bgr_device": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                               C.c_double]),
    "dvo_download_level": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvo_slot_info": (C.c_int, [_ctx, C.c_int, _c_int_p, _c_int_p, _c_int_p]),
    "dvo_download_points": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _c_int_p]),
    "dvo_normal_equations": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "dvo_weighted_r2_sum": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "dvo_weighted_r2_stats": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "dvo_residuals_jacobian": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvo_solve_batch": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int,
                                  C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "dvo_solve_batch_enqueue": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                          C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "dvo_solve_batch_fetch": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvo_debug_divide": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "dvo_set_prior": (C.c_int, [_ctx, C.c_double, C.c_void_p, C.c_int]),
    "dvo_set_level_iterations": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "dvo_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dvo_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dvo_allgather_poses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dvo_comm_destroy": (None, [C.c_void_p]),
    "dvo_comm_last_error": (C.c_char_p, []),
    "dvo_trace_enable": (C.c_int, [_ctx, C.c_int]),
    "dvo_student_t_fallbacks": (C.c_int, [_ctx, _c_i64_p]),
    "dvo_student_t_fallback_pairs": (C.c_int, [_ctx, _c_i64_p]),
    "dvo_trace_read": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "dvo_profile_enable": (C.c_int, [_ctx, C.c_int]),
    "dvo_profile_read": (C.c_int, [_ctx, _c_i64_p, _c_double_p, C.c_int]),
    "dvo_debug_valu_rate": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, _c_double_p]),
    "dvo_timer_start": (C.c_int, [_ctx]),
    "dvo_timer_stop": (C.c_int, [_ctx, _c_float_p]),
}


# entry points of experiment builds only (-DDVO_EXPERIMENTS, include/dvo_hip.h): bound when the loaded library has them
EXPERIMENT_SIGNATURES = {
    "dvo_debug_small_stamps": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_int]),
}


class DVOHipError(RuntimeError):
    """A C-ABI call returned a negative dvo_status."""

    def __init__(self, code: int, message: str):
        super().__init__("libdvo_hip error {}: {}".format(code, message))
        self.code = code


def library_path() -> Path:
    return _LIB_PATH


def load_library() -> C.CDLL:
    if not _LIB_PATH.exists():
        raise ImportError(
            "HIP extension {} is missing — build it with `python __graft_entry__.py` or "
            "`make -C dense-visual-odometry_amd/csrc`; this package has no CPU fallback".format(_LIB_PATH))
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and loads it by path, so when libdvo_hip.so (linked
    # against the system ROCm) comes first the process ends up with two runtimes and torch.cuda then reports "No HIP GPUs are
    # available" (measured, tools/hip_load_order.py).  With torch imported first the loader resolves this library's libamdhip64
    # dependency to the copy already in the process.  Hosts without PyTorch (plain C, or Python without torch) are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENT_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    return lib


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _as(a, dtype, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None:
        out = out.reshape(shape)
    return out


class HipContext:
    """Owns one `dvo_ctx` (one per GPU).  Thin, typed access to the C ABI with NumPy in/out."""

    def __init__(self, device: int = 0, max_w: int = 640, max_h: int = 480, max_levels: int = 4, n_slots: int = 2,
                 max_pairs: int = 1):
        self._lib = lib()
        self._h = _ctx()
        rc = self._lib.dvo_create(device, max_w, max_h, max_levels, n_slots, max_pairs, C.byref(self._h))
        if rc != 0:
            msg = self._lib.dvo_last_error(None)
            raise DVOHipError(rc, msg.decode() if msg else "dvo_create failed")
        self.device, self.max_w, self.max_h = device, max_w, max_h
        self.max_levels, self.n_slots, self.max_pairs = max_levels, n_slots, max_pairs

    # -- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.dvo_last_error(self._h)
            raise DVOHipError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dvo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream: Optional[int]):
        self._check(self._lib.dvo_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        self._check(self._lib.dvo_synchronize(self._h))

    def wait_for(self, other: "HipContext"):
        """Work enqueued on this ctx from now on waits (on the GPU) for everything enqueued on `other` so far."""
        self._check(self._lib.dvo_wait_for(self._h, other._h))

    def set_upload_peer(self, peer: Optional["HipContext"]):
        """Host->device copies of `upload_frames` go to `peer`'s in-order copy stream (None: back to this ctx's stream)."""
        self._check(self._lib.dvo_set_upload_peer(self._h, peer._h if peer is not None else None))

    # -- camera -----------------------------------------------------------------------------------
    def set_camera(self, fx, fy, cx, cy, skew, depth_scale):
        self._check(self._lib.dvo_set_camera(self._h, float(fx), float(fy), float(cx), float(cy), float(skew),
                                             float(depth_scale)))

    def level_intrinsics(self, level: int) -> np.ndarray:
        k = np.zeros((3, 4), dtype=np.float32)
        self._check(self._lib.dvo_get_level_intrinsics(self._h, level, k.ctypes.data_as(_c_float_p)))
        return k

    # -- frames -----------------------------------------------------------------------------------
    def upload_frame(self, slot: int, gray: np.ndarray, depth: np.ndarray, levels: int):
        g = _as(gray, np.uint8)
        d = _as(depth, np.uint16)
        if g.ndim != 2 or g.shape != d.shape:
            raise ValueError("gray {} and depth {} must be 2-D arrays of one shape".format(g.shape, d.shape))
        h, w = g.shape
        self._check(self._lib.dvo_upload_frame(self._h, slot, _ptr(g), _ptr(d), w, h, levels))

    def upload_frame_device(self, slot: int, d_gray_ptr: int, d_depth_ptr: int, w: int, h: int, levels: int):
        self._check(self._lib.dvo_upload_frame_device(self._h, slot, C.c_void_p(d_gray_ptr), C.c_void_p(d_depth_ptr),
                                                      w, h, levels))

    def upload_frames_device(self, slots, d_gray_ptr: int, d_depth_ptr: int, w: int, h: int, levels: int):
        """n frames from contiguous [n][h][w] device arrays (u8 gray, u16 depth) into `slots` (async)."""
        sl = _as(slots, np.int32)
        self._check(self._lib.dvo_upload_frames_device(self._h, sl.shape[0], _ptr(sl), C.c_void_p(d_gray_ptr),
                                                       C.c_void_p(d_depth_ptr), w, h, levels))

    def upload_frames(self, slots, gray: np.ndarray, depth: np.ndarray, levels: int):
        """n frames from contiguous host arrays [n][h][w] (u8 gray, u16 depth).  Asynchronous when the arrays are
        page-locked (`pinned_array`): keep them untouched until the next blocking call."""
        sl = _as(slots, np.int32)
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        if g.ndim != 3 or g.shape != d.shape or g.shape[0] != sl.shape[0]:
            raise ValueError("gray/depth must be [n][h][w] arrays matching the slot list")
        self._keepalive = (g, d)
        self._check(self._lib.dvo_upload_frames(self._h, sl.shape[0], _ptr(sl), _ptr(g), _ptr(d), g.shape[2], g.shape[1], levels))

    def upload_frames_bgr(self, slots, bgr: np.ndarray, depth: np.ndarray, levels: int, max_distance: float = 5.0):
        """n colour frames from contiguous host arrays [n][h][w][3] (BGR u8) / [n][h][w] (u16 depth): two large copies, BGR->gray
        and the depth clamp on the device.  Asynchronous for page-locked arrays (`pinned_array`)."""
        sl = _as(slots, np.int32)
        c = np.ascontiguousarray(bgr, dtype=np.uint8)
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        if c.ndim != 4 or c.shape[3] != 3 or c.shape[:3] != d.shape or c.shape[0] != sl.shape[0]:
            raise ValueError("bgr must be [n][h][w][3] and depth [n][h][w], matching the slot list")
        self._keepalive = (c, d)
        self._check(self._lib.dvo_upload_frames_bgr(self._h, sl.shape[0], _ptr(sl), _ptr(c), _ptr(d), c.shape[2], c.shape[1],
                                                    levels, float(max_distance)))

    def upload_frames_bgr_device(self, slots, d_bgr_ptr: int, d_depth_ptr: int, w: int, h: int, levels: int,
                                 max_distance: float = 5.0):
        sl = _as(slots, np.int32)
        self._check(self._lib.dvo_upload_frames_bgr_device(self._h, sl.shape[0], _ptr(sl), C.c_void_p(d_bgr_ptr),
                                                           C.c_void_p(d_depth_ptr), w, h, levels, float(max_distance)))

    def upload_frame_bgr(self, slot: int, bgr: np.ndarray, depth: np.ndarray, levels: int, max_distance: float):
        c = _as(bgr, np.uint8)
        d = _as(depth, np.uint16)
        if c.ndim != 3 or c.shape[2] != 3 or c.shape[:2] != d.shape:
            raise ValueError("bgr {} must be HxWx3 matching depth {}".format(c.shape, d.shape))
        h, w = d.shape
        self._check(self._lib.dvo_upload_frame_bgr(self._h, slot, _ptr(c), _ptr(d), w, h, levels, float(max_distance)))

    def slot_info(self, slot: int):
        w, h, lv = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.dvo_slot_info(self._h, slot, C.byref(w), C.byref(h), C.byref(lv)))
        return w.value, h.value, lv.value

    def download_points(self, slot: int, level: int):
        """Valid-point list of one level -> dict(x, y, intensity, z) in raster order (depth-valid pixels only)."""
        n = C.c_int(0)
        self._check(self._lib.dvo_download_points(self._h, slot, level, None, None, 0, C.byref(n)))
        z = np.empty(n.value, dtype=np.float32)
        pix = np.empty(n.value, dtype=np.uint32)
        self._check(self._lib.dvo_download_points(self._h, slot, level, _ptr(z), _ptr(pix), n.value, C.byref(n)))
        return {"x": (pix & 0xfff).astype(np.int32), "y": ((pix >> 12) & 0xfff).astype(np.int32),
                "intensity": (pix >> 24).astype(np.uint8), "z": z}

    def download_level(self, slot: int, level: int, want=("gray", "depth", "gx", "gy")):
        w, h, lv = self.slot_info(slot)
        if lv == 0:
            raise DVOHipError(-5, "slot {} holds no frame".format(slot))
        for _ in range(level):
            w, h = (w + 1) // 2, (h + 1) // 2
        out = {}
        if "gray" in want:
            out["gray"] = np.empty((h, w), dtype=np.uint8)
        if "depth" in want:
            out["depth"] = np.empty((h, w), dtype=np.uint16)
        if "gx" in want:
            out["gx"] = np.empty((h, w), dtype=np.int16)
        if "gy" in want:
            out["gy"] = np.empty((h, w), dtype=np.int16)
        self._check(self._lib.dvo_download_level(self._h, slot, level, _ptr(out.get("gray")), _ptr(out.get("depth")),
                                                 _ptr(out.get("gx")), _ptr(out.get("gy"))))
        return out

    # -- hot path ---------------------------------------------------------------------------------
    def normal_equations(self, level: int, prev_slots: Sequence[int], cur_slots: Sequence[int], T: np.ndarray,
                         variant: int = VARIANT_V1, weight_mode: int = WEIGHT_NONE,
                         weight_param: Optional[np.ndarray] = None) -> np.ndarray:
        """-> (n_pairs, 32) float64 records (layout in include/dvo_hip.h)."""
        ps = _as(prev_slots, np.int32)
        cs = _as(cur_slots, np.int32)
        n = ps.shape[0]
        Tm = _as(T, np.float32).reshape(n, -1)
        if Tm.shape[1] == 16:
            Tm = np.ascontiguousarray(Tm[:, :12])
        if Tm.shape[1] != 12:
            raise ValueError("T must be n x 3x4 (or 4x4) row-major")
        wp = None if weight_param is None else _as(weight_param, np.float64, (n,))
        out = np.empty((n, NE_STRIDE), dtype=np.float64)
        self._check(self._lib.dvo_normal_equations(self._h, level, n, _ptr(ps), _ptr(cs), _ptr(Tm), variant,
                                                   weight_mode, _ptr(wp), _ptr(out)))
        return out

    def weighted_r2_sum(self, level: int, prev_slots, cur_slots, T, lam, variant: int = VARIANT_V1) -> np.ndarray:
        ps = _as(prev_slots, np.int32)
        cs = _as(cur_slots, np.int32)
        n = ps.shape[0]
        Tm = _as(T, np.float32).reshape(n, -1)
        if Tm.shape[1] == 16:
            Tm = np.ascontiguousarray(Tm[:, :12])
        lm = _as(lam, np.float64, (n,))
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.dvo_weighted_r2_sum(self._h, level, n, _ptr(ps), _ptr(cs), _ptr(Tm), variant, _ptr(lm),
                                                  _ptr(out)))
        return out

    def weighted_r2_stats(self, level: int, prev_slots, cur_slots, T, lam, variant: int = VARIANT_V1) -> np.ndarray:
        """-> (n_pairs, 2): the Student-t scale sum and the number of residuals it ran over."""
        ps = _as(prev_slots, np.int32)
        cs = _as(cur_slots, np.int32)
        n = ps.shape[0]
        Tm = _as(T, np.float32).reshape(n, -1)
        if Tm.shape[1] == 16:
            Tm = np.ascontiguousarray(Tm[:, :12])
        lm = _as(lam, np.float64, (n,))
        out = np.empty((n, 2), dtype=np.float64)
        self._check(self._lib.dvo_weighted_r2_stats(self._h, level, n, _ptr(ps), _ptr(cs), _ptr(Tm), variant, _ptr(lm),
                                                    _ptr(out)))
        return out

    def residuals_jacobian(self, level: int, prev_slot: int, cur_slot: int, T: np.ndarray, variant: int = VARIANT_V1,
                           want_uv: bool = True):
        """Dense (h*w) outputs: r, J (h*w,6), valid (bool), uv (h*w,2)."""
        w, h, lv = self.slot_info(prev_slot)
        for _ in range(level):
            w, h = (w + 1) // 2, (h + 1) // 2
        n = w * h
        Tm = _as(T, np.float32).reshape(-1)[:12].copy()
        r = np.empty(n, dtype=np.float32)
        J = np.empty((n, 6), dtype=np.float32)
        valid = np.empty(n, dtype=np.uint8)
        uv = np.empty((n, 2), dtype=np.float32) if want_uv else None
        self._check(self._lib.dvo_residuals_jacobian(self._h, level, prev_slot, cur_slot, _ptr(Tm), variant, _ptr(r),
                                                     _ptr(J), _ptr(valid), _ptr(uv)))
        return r, J, valid.astype(bool), uv, (h, w)

    def solve_batch(self, prev_slots, cur_slots, levels: int, max_iterations: int = 100, tolerance: float = 1e-6,
                    max_increased_steps_allowed: int = 0, variant: int = VARIANT_V1, weight_mode: int = WEIGHT_NONE,
                    weight_param: float = 0.0, T_init: Optional[np.ndarray] = None):
        """Whole coarse-to-fine GN solve on the device. -> dict(T (n,3,4) f32, xi (n,6) f64, iters, err)."""
        self.solve_batch_enqueue(prev_slots, cur_slots, levels, max_iterations, tolerance,
                                 max_increased_steps_allowed, variant, weight_mode, weight_param, T_init)
        return self.solve_batch_fetch(len(prev_slots))

    def solve_batch_enqueue(self, prev_slots, cur_slots, levels, max_iterations=100, tolerance=1e-6,
                            max_increased_steps_allowed=0, variant=VARIANT_V1, weight_mode=WEIGHT_NONE,
                            weight_param=0.0, T_init=None):
        ps = _as(prev_slots, np.int32)
        cs = _as(cur_slots, np.int32)
        n = ps.shape[0]
        Ti = None
        if T_init is not None:
            Ti = _as(T_init, np.float32).reshape(n, -1)
            if Ti.shape[1] == 16:
                Ti = np.ascontiguousarray(Ti[:, :12])
        self._check(self._lib.dvo_solve_batch_enqueue(self._h, n, _ptr(ps), _ptr(cs), levels, max_iterations,
                                                      float(tolerance), max_increased_steps_allowed, variant,
                                                      weight_mode, float(weight_param), _ptr(Ti)))

    def solve_batch_fetch(self, n: int):
        T = np.empty((n, 3, 4), dtype=np.float32)
        xi = np.empty((n, 6), dtype=np.float64)
        iters = np.empty(n, dtype=np.int32)
        err = np.empty(n, dtype=np.float64)
        self._check(self._lib.dvo_solve_batch_fetch(self._h, n, _ptr(T), _ptr(xi), _ptr(iters), _ptr(err)))
        return dict(T=T, xi=xi, iters=iters, err=err)

    def set_level_iterations(self, iters) -> None:
        """Per-level iteration schedule (level 0 = finest) for the following solve_batch calls; None restores one max_iterations
        for every level (the reference's behaviour)."""
        if iters is None:
            self._check(self._lib.dvo_set_level_iterations(self._h, None, 0))
            return
        a = np.ascontiguousarray(np.asarray(iters, dtype=np.int32).reshape(-1))
        self._check(self._lib.dvo_set_level_iterations(self._h, _ptr(a), int(a.shape[0])))

    def set_prior(self, sigma: Optional[float], last_qt: Optional[np.ndarray] = None):
        """Motion prior for the following solve_batch calls (sigma None/<=0: off). last_qt: (n,7) qw qx qy qz tx ty tz."""
        if sigma is None or not sigma > 0:
            self._check(self._lib.dvo_set_prior(self._h, 0.0, None, 0))
            return
        q = None if last_qt is None else _as(last_qt, np.float64).reshape(-1, 7)
        self._check(self._lib.dvo_set_prior(self._h, float(sigma), _ptr(q), 0 if q is None else q.shape[0]))

    def debug_divide(self, a, b: np.ndarray):
        """(kernel quotient, IEEE quotient) computed on the device for float32 arrays a, b; a = None: (the projection's refined
        reciprocal of b, the IEEE 1 / b)."""
        b = _as(b, np.float32).reshape(-1)
        a = None if a is None else _as(a, np.float32).reshape(-1)
        out = np.empty_like(b)
        ref = np.empty_like(b)
        self._check(self._lib.dvo_debug_divide(self._h, _ptr(a), _ptr(b), _ptr(out), _ptr(ref), b.shape[0]))
        return out, ref

    def trace_enable(self, cap: int):
        self._check(self._lib.dvo_trace_enable(self._h, cap))
        self._trace_cap = cap

    def trace_read(self, n: int) -> np.ndarray:
        out = np.zeros((n, self._trace_cap, TRACE_STRIDE), dtype=np.float64)
        self._check(self._lib.dvo_trace_read(self._h, n, _ptr(out)))
        return out

    # -- measurement ------------------------------------------------------------------------------
    def profile_enable(self, on: bool):
        self._check(self._lib.dvo_profile_enable(self._h, int(bool(on))))

    def profile_read(self, reset: bool = True):
        launches = np.zeros(MAX_LEVELS, dtype=np.int64)
        ms = np.zeros(MAX_LEVELS, dtype=np.float64)
        self._check(self._lib.dvo_profile_read(self._h, launches.ctypes.data_as(_c_i64_p),
                                               ms.ctypes.data_as(_c_double_p), int(bool(reset))))
        return launches, ms

    def student_t_fallbacks(self) -> int:
        """Solves with the reference's Student-t that had to be repeated with the three-pass form (include/dvo_hip.h)."""
        n = C.c_int64()
        self._check(self._lib.dvo_student_t_fallbacks(self._h, C.byref(n)))
        return int(n.value)

    def student_t_fallback_pairs(self) -> int:
        """Pairs this context solved again because a one-pass Student-t iteration could not be bounded or the IRLS had another shape."""
        n = C.c_int64()
        self._check(self._lib.dvo_student_t_fallback_pairs(self._h, C.byref(n)))
        return int(n.value)

    def small_stamps(self, n_pairs: int, reset: bool = True) -> np.ndarray:
        """(n_pairs, 8) uint64 phase clocks of the persistent small-level kernel (experiment build of the library, -DDVO_EXPERIMENTS,
        and a ctx created under DVO_SMALL_STAMPS)."""
        if not hasattr(self._lib, "dvo_debug_small_stamps"):
            raise DVOHipError(-4, "dvo_debug_small_stamps: the loaded library is not an experiment build (-DDVO_EXPERIMENTS)")
        out = np.zeros((n_pairs, 8), dtype=np.uint64)
        self._check(self._lib.dvo_debug_small_stamps(self._h, n_pairs, _ptr(out), int(bool(reset))))
        return out

    def valu_rate(self, waves_per_simd: int = 5, iterations: int = 4000) -> dict:
        """ns per VALU wave-instruction per SIMD at `waves_per_simd` resident waves: for the composition of one trip of the level-0
        reduction's record loop ("composition": simple, simple with a VGPR bank conflict, conversion-class, v_rcp_f32 instructions)
        and for each class alone."""
        out = np.zeros(128, dtype=np.float64)       # the product library writes 6 values; an experiments build (DVO_HIP_LIB) one per probe mode
        comp = np.zeros(4, dtype=np.int32)
        self._check(self._lib.dvo_debug_valu_rate(self._h, waves_per_simd, iterations, _ptr(comp), out.ctypes.data_as(_c_double_p)))
        f, c, s, r = (int(v) for v in comp)
        priced = (f * out[1] + c * out[2] + s * out[3] + r * out[4]) / max(f + c + s + r, 1)
        return {"mix": float(out[0]), "fast": float(out[1]), "conflict": float(out[2]), "slow": float(out[3]), "rcp": float(out[4]),
                "mix_small_blocks": float(out[5]), "mix_class_priced": float(priced), "experiment_orders": [float(out[6]), float(out[7])],
                "waves_per_simd": waves_per_simd, "composition": [f, c, s, r]}

    def timer_start(self):
        self._check(self._lib.dvo_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self._lib.dvo_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)


class RcclPoseGather:
    """RCCL all-gather of pose records through the C ABI (dvo_comm_*): for hosts without torch.distributed.
    Rank 0 calls `RcclPoseGather.unique_id()` and ships the 128 bytes to the other ranks (file, socket, MPI...)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = lib().dvo_comm_unique_id(buf)
        if rc != 0:
            raise DVOHipError(rc, (lib().dvo_comm_last_error() or b"").decode())
        return buf.raw

    def __init__(self, device: int, rank: int, nranks: int, unique_id: bytes):
        self._lib = lib()
        self._h = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        rc = self._lib.dvo_comm_create(device, rank, nranks, C.c_char_p(unique_id), C.byref(self._h))
        if rc != 0:
            raise DVOHipError(rc, (self._lib.dvo_comm_last_error() or b"").decode())

    def allgather(self, records: np.ndarray) -> np.ndarray:
        """(n,16) float32 records of this rank -> (nranks*n, 16), rank-major (n must be equal on all ranks)."""
        rec = _as(records, np.float32).reshape(-1, 16)
        out = np.empty((self.nranks * rec.shape[0], 16), dtype=np.float32)
        rc = self._lib.dvo_allgather_poses(self._h, _ptr(rec), rec.shape[0], _ptr(out))
        if rc != 0:
            raise DVOHipError(rc, (self._lib.dvo_comm_last_error() or b"").decode())
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dvo_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PinnedBuffer:
    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        rc = lib().dvo_host_alloc(nbytes, C.byref(self.ptr))
        if rc != 0:
            raise DVOHipError(rc, "dvo_host_alloc failed")
        self.nbytes = nbytes

    def __del__(self):
        try:
            if self.ptr:
                lib().dvo_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_array(shape, dtype) -> np.ndarray:
    """NumPy array backed by page-locked host memory (dvo_host_alloc); freed when the array is collected."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    buf = _PinnedBuffer(max(n, 1))
    raw = (C.c_uint8 * max(n, 1)).from_address(buf.ptr.value)
    arr = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    arr.flags.writeable = True
    _PINNED[id(raw)] = buf                     # tie the allocation's lifetime to the ctypes buffer
    import weakref
    weakref.finalize(raw, _PINNED.pop, id(raw), None)
    return arr


_PINNED = {}


def unpack_record(rec: np.ndarray):
    """(32,) record -> H (6,6), b (6,) [already negated: b = -J^T W r], sum_wr2, n, sum_r2."""
    rec = np.asarray(rec, dtype=np.float64)
    H = rec[_FULL_FROM_TRIU]                       # (6,6) gather of the 21 upper-triangle entries
    return H, -rec[21:27], float(rec[27]), int(round(rec[28])), float(rec[29])


def _full_from_triu():
    idx = np.zeros((6, 6), dtype=np.intp)
    q = 0
    for i in range(6):
        for j in range(i, 6):
            idx[i, j] = idx[j, i] = q
            q += 1
    return idx


_FULL_FROM_TRIU = _full_from_triu()
