"""Batched / sharded front end of the HIP core: many frame pairs per launch, one process per GPU.

With the reference's defaults (`init_guess = identity`, core/base_dense_visual_odometry.py:56, and
`sigma=None`, base_robust_dvo.py:36) the transform of frame pair (t-1, t) depends on those two frames
only, so the pairs of a sequence — and separate sequences — are independent units of work
(SURVEY 8e).  `BatchAligner` keeps a window of frames resident on one GPU and aligns all its
consecutive pairs in one device-side Gauss-Newton solve; `shard_pairs` / `allgather_poses` split a
sequence over ranks and gather the per-pair pose records (16 floats each) with one RCCL all-gather;
`compose_trajectory` is the sequential pose chaining `pose_t = pose_{t-1} * T_t^-1` (:79) on the host.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

from . import _cabi

_WEIGHTS = {"none": _cabi.WEIGHT_NONE, "student_t_reference": _cabi.WEIGHT_STUDENT_T, "huber": _cabi.WEIGHT_HUBER,
            "student_t": _cabi.WEIGHT_STUDENT_T_MEAN}


class BatchAligner:
    """Aligns up to `max_pairs` consecutive frame pairs per call on one GPU."""

    def __init__(self, K: np.ndarray, depth_scale: float, width: int = 640, height: int = 480, levels: int = 4,
                 max_pairs: int = 64, device: int = 0, max_iterations: int = 10, tolerance: float = 1e-6,
                 max_increased_steps_allowed: int = 0, weighting: str = "none", huber_k: float = 1.345,
                 approximate_image2_gradient: bool = False, reference_variant: str = "cpu", level_iterations=None):
        if weighting not in _WEIGHTS:
            raise ValueError("weighting must be one of {}".format(sorted(_WEIGHTS)))
        self.width, self.height, self.levels, self.max_pairs = width, height, levels, max_pairs
        self.max_iterations, self.tolerance = max_iterations, tolerance
        self.max_increased = max_increased_steps_allowed
        self.weight_mode, self.huber_k = _WEIGHTS[weighting], huber_k
        self.variant = _cabi.VARIANT_V2 if approximate_image2_gradient else _cabi.VARIANT_V1
        if reference_variant not in ("cpu", "cuda", "gn"):
            raise ValueError("reference_variant must be 'cpu', 'cuda' or 'gn', got '{}'".format(reference_variant))
        if reference_variant == "cuda":
            self.variant = _cabi.VARIANT_V3
        elif reference_variant == "gn":         # build-defined true Gauss-Newton mode (DVO_VARIANT_GN)
            self.variant = _cabi.VARIANT_GN
        self.ctx = _cabi.HipContext(device=device, max_w=width, max_h=height, max_levels=levels,
                                    n_slots=max_pairs + 1, max_pairs=max_pairs)
        K = np.asarray(K, dtype=np.float32)
        self.ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], depth_scale)
        self.ctx.set_level_iterations(level_iterations)      # build-defined schedule (level 0 = finest); None = max_iterations everywhere
        self._slots = np.arange(max_pairs + 1, dtype=np.int32)

    def close(self):
        self.ctx.close()

    # -- frames already resident in HBM (torch tensors / raw device pointers) ---------------------------
    def ingest_device(self, d_gray_ptr: int, d_depth_ptr: int, n_frames: int):
        """n_frames contiguous [n][h][w] images (u8 gray, u16 depth) -> slots 0..n-1: pyramids + packed
        Sobel planes built on the device (async)."""
        if not (2 <= n_frames <= self.max_pairs + 1):
            raise ValueError("n_frames must be in [2, {}]".format(self.max_pairs + 1))
        self.ctx.upload_frames_device(self._slots[:n_frames], d_gray_ptr, d_depth_ptr, self.width, self.height,
                                      self.levels)

    def ingest_host(self, gray: np.ndarray, depth: np.ndarray):
        """[n][h][w] host arrays -> slots 0..n-1 in one batched upload (asynchronous for page-locked arrays, see
        `_cabi.pinned_array`)."""
        n = gray.shape[0]
        if not (2 <= n <= self.max_pairs + 1):
            raise ValueError("n_frames must be in [2, {}]".format(self.max_pairs + 1))
        self.ctx.upload_frames(self._slots[:n], gray, depth, self.levels)

    def ingest_host_bgr(self, bgr: np.ndarray, depth: np.ndarray, max_distance: float = 5.0):
        """[n][h][w][3] BGR + [n][h][w] depth host arrays -> slots 0..n-1: batched upload, colour conversion and depth clamp
        (core/base_dense_visual_odometry.py:58-59) on the device."""
        n = bgr.shape[0]
        if not (2 <= n <= self.max_pairs + 1):
            raise ValueError("n_frames must be in [2, {}]".format(self.max_pairs + 1))
        self.ctx.upload_frames_bgr(self._slots[:n], bgr, depth, self.levels, max_distance)

    def solve_enqueue(self, n_frames: int, T_init: Optional[np.ndarray] = None):
        prev = self._slots[:n_frames - 1]
        self.ctx.solve_batch_enqueue(prev, prev + 1, self.levels, self.max_iterations, self.tolerance,
                                     self.max_increased, self.variant, self.weight_mode, self.huber_k, T_init)

    def solve_fetch(self, n_frames: int):
        return self.ctx.solve_batch_fetch(n_frames - 1)

    def align_device(self, d_gray_ptr: int, d_depth_ptr: int, n_frames: int):
        """ingest + solve of the n_frames-1 consecutive pairs. -> dict(T, xi, iters, err)."""
        self.ingest_device(d_gray_ptr, d_depth_ptr, n_frames)
        self.solve_enqueue(n_frames)
        return self.solve_fetch(n_frames)

    def align_host(self, gray: np.ndarray, depth: np.ndarray):
        self.ingest_host(gray, depth)
        self.solve_enqueue(gray.shape[0])
        return self.solve_fetch(gray.shape[0])


class StreamedAligner:
    """`n_streams` BatchAligners (one HIP stream each) on one GPU working on disjoint slices of a window of
    frames.  The GN loop of a slice is a chain of dependent launches (reduce -> solve -> reduce ...); with two or
    more independent chains in flight the GPU fills the bubbles of one chain (tiny solve kernels, launch gaps,
    the tail of each reduction) with work of another."""

    def __init__(self, K, depth_scale, width=640, height=480, levels=4, max_pairs=64, n_streams=2, device=0, **kw):
        if n_streams < 1 or max_pairs < n_streams:
            raise ValueError("need 1 <= n_streams <= max_pairs")
        self.n_streams, self.max_pairs = n_streams, max_pairs
        self.width, self.height = width, height
        per = (max_pairs + n_streams - 1) // n_streams
        self.parts = [BatchAligner(K, depth_scale, width, height, levels, max_pairs=per, device=device, **kw)
                      for _ in range(n_streams)]
        self.ctx = self.parts[0].ctx
        self._inflight = None                      # slices of the window submit_device left on the GPU
        for a in self.parts:                       # host uploads of all slices take turns on one in-order copy stream
            a.ctx.set_upload_peer(self.ctx)

    def close(self):
        if self._inflight is not None:             # a window still in flight: wait for it before its buffers go away
            self._inflight = None
            self.synchronize()
        for a in reversed(self.parts):             # parts[0] owns the shared copy stream: closed last
            a.close()

    def _slices(self, n_frames):
        n_pairs = n_frames - 1
        out = []
        for s in range(self.n_streams):
            lo, hi = shard_pairs(n_pairs, s, self.n_streams)
            if hi > lo:
                out.append((self.parts[s], lo, hi))
        return out

    def align_device(self, d_gray_ptr: int, d_depth_ptr: int, n_frames: int):
        """Frames [n][h][w] resident in HBM (u8 gray, u16 depth) -> dict(T, xi, iters, err) for the n-1 pairs."""
        npx = self.width * self.height
        sl = self._slices(n_frames)
        for a, lo, hi in sl:                                   # enqueue everything first, then fetch
            a.ingest_device(d_gray_ptr + lo * npx, d_depth_ptr + 2 * lo * npx, hi - lo + 1)
            a.solve_enqueue(hi - lo + 1)
        res = [a.solve_fetch(hi - lo + 1) for a, lo, hi in sl]
        return {k: np.concatenate([r[k] for r in res], axis=0) for k in res[0]}

    def submit_device(self, d_gray_ptr: int, d_depth_ptr: int, n_frames: int):
        """Pipelined form of align_device for a stream of windows: slice by slice, the result of the slice's PREVIOUS window is
        fetched and its next window (ingest + solve) enqueued right behind it, so the GPU never drains between windows (a window
        that ends with a fetch of every slice leaves it idle while the host collects four results and enqueues the next
        launches: ~0.3 ms of an 11.4 ms VGA window of 1024 pairs).  Returns the previous window's result (None for the first call);
        drain() returns the last one.  Windows in flight must not change size between submit_device calls without a drain()."""
        npx = self.width * self.height
        sl = self._slices(n_frames)
        prev = self._inflight
        if prev is not None and [(lo, hi) for _, lo, hi in prev] != [(lo, hi) for _, lo, hi in sl]:
            raise ValueError("window size changed with a window in flight: call drain() first")
        res = []
        for a, lo, hi in sl:
            if prev is not None:
                res.append(a.solve_fetch(hi - lo + 1))
            a.ingest_device(d_gray_ptr + lo * npx, d_depth_ptr + 2 * lo * npx, hi - lo + 1)
            a.solve_enqueue(hi - lo + 1)
        self._inflight = sl
        return {k: np.concatenate([r[k] for r in res], axis=0) for k in res[0]} if res else None

    def drain(self):
        """Result of the window submit_device left in flight (None if there is none)."""
        sl, self._inflight = self._inflight, None
        if sl is None:
            return None
        res = [a.solve_fetch(hi - lo + 1) for a, lo, hi in sl]
        return {k: np.concatenate([r[k] for r in res], axis=0) for k in res[0]}

    def submit_host(self, img: np.ndarray, depth: np.ndarray, colour: bool = False, max_distance: float = 5.0):
        """Pipelined form of align_host / align_host_bgr for a stream of windows in (page-locked) host memory, as submit_device: slice
        by slice, the result of the slice's previous window is fetched and its next window (upload + ingest + solve) enqueued right
        behind it.  The arrays must stay valid until the NEXT submit_host / drain has returned (their uploads are asynchronous).
        A window of another size than the one in flight is preceded by a drain.  Returns the previous window's result (None for the
        first call); drain() returns the last one."""
        n_frames = img.shape[0]
        sl = self._slices(n_frames)
        prev = self._inflight
        drained = None
        if prev is not None and [(lo, hi) for _, lo, hi in prev] != [(lo, hi) for _, lo, hi in sl]:
            drained, prev = self.drain(), None
        res = []
        for a, lo, hi in sl:
            if prev is not None:
                res.append(a.solve_fetch(hi - lo + 1))
            if colour:
                a.ingest_host_bgr(img[lo:hi + 1], depth[lo:hi + 1], max_distance)
            else:
                a.ingest_host(img[lo:hi + 1], depth[lo:hi + 1])
            a.solve_enqueue(hi - lo + 1)
        self._inflight = sl
        if drained is not None:
            return drained
        return {k: np.concatenate([r[k] for r in res], axis=0) for k in res[0]} if res else None

    def align_host(self, gray: np.ndarray, depth: np.ndarray):
        """Host frames [n][h][w] (ideally page-locked): the H2D copy of one slice overlaps the solve of another."""
        n_frames = gray.shape[0]
        sl = self._slices(n_frames)
        for a, lo, hi in sl:                    # copies queue up on the shared copy stream in this order; the kernels
            a.ingest_host(gray[lo:hi + 1], depth[lo:hi + 1])      # of a slice start behind its own copies only
            a.solve_enqueue(hi - lo + 1)
        res = [a.solve_fetch(hi - lo + 1) for a, lo, hi in sl]
        return {k: np.concatenate([r[k] for r in res], axis=0) for k in res[0]}

    def align_host_bgr(self, bgr: np.ndarray, depth: np.ndarray, max_distance: float = 5.0):
        """Colour host frames [n][h][w][3] + depth [n][h][w] (ideally page-locked): as align_host, conversion on the device."""
        n_frames = bgr.shape[0]
        sl = self._slices(n_frames)
        for a, lo, hi in sl:
            a.ingest_host_bgr(bgr[lo:hi + 1], depth[lo:hi + 1], max_distance)
            a.solve_enqueue(hi - lo + 1)
        res = [a.solve_fetch(hi - lo + 1) for a, lo, hi in sl]
        return {k: np.concatenate([r[k] for r in res], axis=0) for k in res[0]}

    def synchronize(self):
        for a in self.parts:
            a.ctx.synchronize()


# -- sharding over GPUs ------------------------------------------------------------------------------
def shard_pairs(n_pairs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of pair indices owned by `rank` (pair i = frames (i, i+1); a rank needs
    frames lo..hi, i.e. one boundary frame is duplicated between neighbouring ranks)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(n_pairs, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pose_records(T: np.ndarray, iters: Optional[np.ndarray] = None, err: Optional[np.ndarray] = None) -> np.ndarray:
    """(n,3,4) transforms -> (n,16) float32 records: 12 floats [R|t], iterations, final error, 2 reserved."""
    n = T.shape[0]
    rec = np.zeros((n, 16), dtype=np.float32)
    rec[:, :12] = np.asarray(T, dtype=np.float32).reshape(n, 12)
    if iters is not None:
        rec[:, 12] = iters
    if err is not None:
        rec[:, 13] = err
    return rec


def allgather_poses(local_records: np.ndarray, counts: Sequence[int], device=None) -> np.ndarray:
    """All-gather of the per-pair pose records over the default process group (RCCL over xGMI when the
    backend is "nccl" on ROCm; gloo on CPU in the tests).  `counts[r]` = records owned by rank r; ranks may
    own different counts, so records are padded to max(counts).  64 B per pair: latency-bound."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local_records, dtype=np.float32)
    world = dist.get_world_size()
    assert len(counts) == world and local_records.shape[0] == counts[dist.get_rank()]
    m = max(counts)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    buf = torch.zeros((m, 16), dtype=torch.float32, device=dev)
    if local_records.shape[0]:
        buf[:local_records.shape[0]] = torch.from_numpy(np.ascontiguousarray(local_records, dtype=np.float32)).to(dev)
    parts = [torch.empty((m, 16), dtype=torch.float32, device=dev) for _ in range(world)]
    dist.all_gather(parts, buf)
    host = torch.stack(parts).cpu().numpy()
    return np.concatenate([host[r, :counts[r]] for r in range(world)], axis=0)


class PoseGatherer:
    """`allgather_poses` for a step loop: the same all-gather of 16-float pose records, with every buffer (page-locked host
    send / receive arrays, device send / receive tensors) allocated once, one `all_gather_into_tensor` per call and
    asynchronous copies around it — a fraction of a millisecond per step does not belong in a 12 ms step.  Padded to
    max(counts) records per rank like `allgather_poses`; works on gloo (CPU tensors) as well."""

    def __init__(self, counts: Sequence[int], device=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.counts = [int(c) for c in counts]
        self.active = dist.is_available() and dist.is_initialized()
        if not self.active:
            return
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        assert len(self.counts) == self.world
        self.m = max(max(self.counts), 1)
        self.dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        on_gpu = str(self.dev).startswith("cuda")
        self.h_send = torch.zeros((self.m, 16), dtype=torch.float32, pin_memory=on_gpu)
        self.h_recv = torch.zeros((self.world * self.m, 16), dtype=torch.float32, pin_memory=on_gpu)
        self.d_send = torch.zeros((self.m, 16), dtype=torch.float32, device=self.dev) if on_gpu else self.h_send
        self.d_recv = torch.zeros((self.world * self.m, 16), dtype=torch.float32, device=self.dev) if on_gpu else self.h_recv
        self._on_gpu = on_gpu
        # The gather's three tiny operations (H2D, the RCCL kernel, D2H) queue behind whatever the GPU is doing; beside four streams of
        # reduction launches that wait is ~2 ms per step on the default stream.  A high-priority stream lets the dispatcher take them
        # first (they occupy one workgroup for microseconds).
        self._stream = torch.cuda.Stream(device=self.dev, priority=-1) if on_gpu else None

    def __call__(self, local_records: np.ndarray) -> np.ndarray:
        if not self.active:
            return np.asarray(local_records, dtype=np.float32)
        n = self.counts[self.rank]
        assert local_records.shape == (n, 16)
        if n:
            self.h_send.numpy()[:n] = local_records
        import contextlib
        with (self._torch.cuda.stream(self._stream) if self._on_gpu else contextlib.nullcontext()):
            if self._on_gpu:
                self.d_send.copy_(self.h_send, non_blocking=True)
            try:
                self._dist.all_gather_into_tensor(self.d_recv, self.d_send)
            except (RuntimeError, NotImplementedError):                # a backend without the flat form
                parts = list(self.d_recv.view(self.world, self.m, 16).unbind(0))
                self._dist.all_gather(parts, self.d_send)
            if self._on_gpu:
                self.h_recv.copy_(self.d_recv, non_blocking=True)
                self._stream.synchronize()
        host = self.h_recv.numpy().reshape(self.world, self.m, 16)
        return np.concatenate([host[r, :self.counts[r]] for r in range(self.world)], axis=0)


def compose_trajectory(T_rel: np.ndarray, init_pose: Optional[np.ndarray] = None) -> np.ndarray:
    """Sequential prefix composition pose_t = pose_{t-1} @ inv(T_t) in float64
    (core/base_dense_visual_odometry.py:79).  T_rel: (n,3,4) or (n,16) records. -> (n+1,4,4)."""
    T_rel = np.asarray(T_rel, dtype=np.float64)
    if T_rel.ndim == 2:
        T_rel = T_rel[:, :12].reshape(-1, 3, 4)
    n = T_rel.shape[0]
    poses = np.empty((n + 1, 4, 4), dtype=np.float64)
    poses[0] = np.eye(4) if init_pose is None else np.asarray(init_pose, dtype=np.float64)
    for i in range(n):
        R, t = T_rel[i, :, :3], T_rel[i, :, 3]
        Tinv = np.eye(4)
        Tinv[:3, :3] = R.T
        Tinv[:3, 3] = -R.T @ t
        poses[i + 1] = poses[i] @ Tinv
    return poses


def absolute_trajectory_error(est_T_wc: np.ndarray, gt_T_wc: np.ndarray) -> float:
    """TUM-style ATE RMSE (m): rigid (Horn / Kabsch) alignment of the estimated camera positions onto the
    ground truth, then RMSE of the position residuals."""
    P = np.asarray(est_T_wc)[:, :3, 3]
    Q = np.asarray(gt_T_wc)[:, :3, 3]
    mp, mq = P.mean(axis=0), Q.mean(axis=0)
    U, _, Vt = np.linalg.svd((Q - mq).T @ (P - mp))
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    res = (Q - mq) - (P - mp) @ R.T
    return float(np.sqrt(np.mean(np.sum(res * res, axis=1))))
