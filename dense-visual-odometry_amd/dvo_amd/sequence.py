"""Sequence front end of the batched aligner: background frame decoding into page-locked windows, windowed batched
alignment, and the report files of the reference's CLI harness.

Counterparts in the reference (src/test_dvo.py): the loaders hold every decoded frame of the sequence in RAM before the
first `step` (:180-191); `main` steps frame by frame (:298-321) and writes `report.json` — lists of the 6-vector logs of
every pose / transform and the per-frame translational errors, plus the loader's additional info — and, for TUM data, the
trajectory as `timestamp tx ty tz qx qy qz qw` (:323-345).  Here decoding runs in a thread one window ahead of the GPU,
a window of frames goes up with one batched upload (`dvo_upload_frames` / `dvo_upload_frames_bgr`) and all its pairs are
aligned in one device-side solve; the report keeps the reference's keys and adds the 4x4 forms and ATE / RPE.
"""
from __future__ import annotations

import json
import queue
import threading
import time
from pathlib import Path
from typing import Callable, Iterable, Iterator, List, Optional, Tuple

import numpy as np

from .lie import Se3
from .batch import compose_trajectory


class Decoded:
    """A frame handed over by a DecoderPool: views into a shared-memory slot, and the call that gives the slot back."""
    __slots__ = ("img", "depth", "release")

    def __init__(self, img, depth, release):
        self.img, self.depth, self.release = img, depth, release


def _decoder_worker(task_q, result_q, shm_name, img_shape, depth_shape, slot_bytes):
    """Worker process of DecoderPool: runs picklable zero-argument decoders, writes (image, depth) into its slot of the shared block."""
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=shm_name)
    n_img = int(np.prod(img_shape))
    try:
        while True:
            task = task_q.get()
            if task is None:
                return
            gen, tid, slot, fn = task
            try:
                img, dep = fn()
                base = slot * slot_bytes
                np.ndarray(img_shape, np.uint8, shm.buf, base)[...] = img
                np.ndarray(depth_shape, np.uint16, shm.buf, base + ((n_img + 63) // 64) * 64)[...] = dep
                result_q.put((gen, tid, slot, None))
            except BaseException as e:                      # surfaced in the parent, in sequence order
                result_q.put((gen, tid, slot, "{}: {}".format(type(e).__name__, e)))
    finally:
        shm.close()


class DecoderPool:
    """`workers` PROCESSES decoding frames (PNG pairs) into a ring of shared-memory slots, results handed out in sequence order.
    Threads stop scaling at ~1.2 k VGA frames/s on this image's PIL (the conversion to an array holds the interpreter lock:
    profiles/r04_sequence_png.txt); processes scale with the cores.  Tasks must be picklable zero-argument callables, e.g.
    functools.partial(tum.read_frame, rgb_path, depth_path, "bgr").

    Create the pool BEFORE the process touches the GPU: the workers are started through multiprocessing's forkserver, which execs
    a fresh interpreter, and a process that has initialised HIP must not exec (nor fork) - the workers themselves never import
    torch or load the HIP library."""

    def __init__(self, workers: int, img_shape, depth_shape, slots: int = 0, method: str = "forkserver"):
        import multiprocessing as mp
        from multiprocessing import shared_memory
        if workers < 1:
            raise ValueError("workers >= 1 required")
        self.img_shape, self.depth_shape = tuple(img_shape), tuple(depth_shape)
        n_img = int(np.prod(self.img_shape))
        self._off_depth = ((n_img + 63) // 64) * 64
        self.slot_bytes = ((self._off_depth + 2 * int(np.prod(self.depth_shape)) + 4095) // 4096) * 4096
        self.slots = slots if slots > 0 else 4 * workers
        self._shm = shared_memory.SharedMemory(create=True, size=self.slots * self.slot_bytes)
        ctx = mp.get_context(method)
        self._tasks, self._results = ctx.Queue(), ctx.Queue()
        self._procs = [ctx.Process(target=_decoder_worker, args=(self._tasks, self._results, self._shm.name, self.img_shape, self.depth_shape,
                                                                 self.slot_bytes), daemon=True) for _ in range(workers)]
        for p in self._procs:
            p.start()
        self._closed = False
        self._gen = 0

    def imap(self, tasks: Iterable) -> Iterator:
        """Yields a Decoded per task, in order; at most `slots` frames are decoded ahead of the consumer.  Items that are not callable
        (already decoded (image, depth) tuples) pass through in place."""
        from collections import deque
        free = deque(range(self.slots))
        self._gen += 1                                     # results of an earlier, abandoned imap are recognised and dropped
        gen = self._gen
        ready, it = {}, iter(tasks)
        submitted = yielded = 0
        exhausted, held = False, None                     # held: a task taken from `tasks` that waits for a free slot
        while True:
            while not exhausted:
                if held is None:
                    try:
                        held = next(it)
                    except StopIteration:
                        exhausted = True
                        break
                if callable(held):
                    if not free:
                        break
                    self._tasks.put((gen, submitted, free.popleft(), held))
                else:
                    ready[submitted] = ("item", held)
                held = None
                submitted += 1
            if yielded == submitted:
                if exhausted:
                    return
                # every slot is with the consumer: it may still be copying the frames out in other threads (FrameWindows does) - wait for
                # a release; a consumer that never releases is an error
                t_wait = time.monotonic()
                while not free:
                    if time.monotonic() - t_wait > 30.0:
                        raise RuntimeError("DecoderPool: every slot is held by the consumer (release a Decoded before asking for the next)")
                    time.sleep(0.0002)
                continue
            while yielded not in ready:
                g, tid, slot, err = self._results.get()
                if g == gen:
                    ready[tid] = ("slot", slot, err)
            r = ready.pop(yielded)
            yielded += 1
            if r[0] == "item":
                yield r[1]
                continue
            _, slot, err = r
            if err is not None:
                free.append(slot)
                raise OSError("frame decoder failed: " + err)
            base = slot * self.slot_bytes
            img = np.ndarray(self.img_shape, np.uint8, self._shm.buf, base)
            dep = np.ndarray(self.depth_shape, np.uint16, self._shm.buf, base + self._off_depth)
            yield Decoded(img, dep, (lambda s=slot: free.append(s)))

    def close(self):
        if self._closed:
            return
        self._closed = True
        for _ in self._procs:
            self._tasks.put(None)
        for p in self._procs:
            p.join(timeout=5.0)
            if p.is_alive():
                p.terminate()
        try:
            self._shm.close()
            self._shm.unlink()
        except (FileNotFoundError, BufferError):
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class FrameWindows:
    """Iterates over windows of consecutive frames of a sequence, decoded by a background thread into a small ring of
    window buffers (page-locked when `alloc` is `_cabi.pinned_array`).  Consecutive windows share one frame: window k holds
    frames [k*W, k*W + W] (W pairs), so every pair of the sequence appears in exactly one window.

    `frames` yields (image, depth) with image HxW (gray u8) or HxWx3 (BGR u8) and depth HxW u16.  Each item of the
    iteration is (image_window, depth_window, n_frames_in_window): views into a ring slot that stay valid until `depth`
    further windows have been requested (`ring` = 3 by default: one being aligned, one uploaded, one being filled)."""

    def __init__(self, frames: Iterable, window_pairs: int, alloc: Callable = None, ring: int = 3, workers=0, copy_threads: int = 4,
                 buffers: Optional[List[Tuple[np.ndarray, np.ndarray]]] = None):
        """`frames` may also yield zero-argument callables returning (image, depth) — e.g. PNG decoders; with `workers` > 0
        they run in a pool of that many threads (PIL releases the GIL while inflating), with `workers` a DecoderPool in its worker
        processes; results are kept in sequence order either way."""
        if window_pairs < 1 or ring < 2:
            raise ValueError("window_pairs >= 1 and ring >= 2 required")
        self._frames = self._resolve(iter(frames), workers)
        self._w = window_pairs
        self._alloc = alloc or (lambda shape, dtype: np.empty(shape, dtype=dtype))
        self._ring = ring
        self._q: "queue.Queue" = queue.Queue(maxsize=ring - 1)
        self._free: "queue.Queue" = queue.Queue()
        self._err: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._fill, name="dvo-frame-decoder", daemon=True)
        # `buffers`: the ring of another FrameWindows of the same geometry (its `.buffers` once it has run) - page-locking a ring of three
        # 129-frame VGA colour windows (600 MB) takes ~0.1 s, which a caller that walks many short sequences need not pay each time
        self._bufs: List[Tuple[np.ndarray, np.ndarray]] = list(buffers) if buffers else []
        if self._bufs:
            if len(self._bufs) != ring or any(b[0].shape[0] != window_pairs + 1 or b[1].shape[0] != window_pairs + 1 for b in self._bufs):
                raise ValueError("buffers must be {} (image, depth) windows of {} frames".format(ring, window_pairs + 1))
            for i in range(ring):
                self._free.put(i)
        self._started = False
        self._stop = False
        self.decode_seconds = 0.0
        # frames are copied into the window by a few threads (NumPy releases the GIL for the copy): one thread moves ~19 k VGA colour
        # frames/s on the GPU box's host, four ~28 k (tools/ubench/copy_rate.py) - one is not enough beside a device that aligns 30 k
        self._copy_threads = max(0, int(copy_threads))

    @staticmethod
    def _resolve(it, workers):
        if isinstance(workers, DecoderPool):
            yield from workers.imap(it)
            return
        if workers <= 0:
            for item in it:
                yield item() if callable(item) else item
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="dvo-png") as pool:
            pending = deque()
            for item in it:
                pending.append(pool.submit(item) if callable(item) else item)
                if len(pending) >= 2 * workers:
                    head = pending.popleft()
                    yield head.result() if hasattr(head, "result") else head
            while pending:
                head = pending.popleft()
                yield head.result() if hasattr(head, "result") else head

    @property
    def buffers(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        """The ring's window arrays (empty until the first frame has arrived), for reuse by a later FrameWindows."""
        return self._bufs

    def _make_buffers(self, img, depth):
        shape = (self._w + 1,) + tuple(img.shape)
        for _ in range(self._ring):
            self._bufs.append((self._alloc(shape, np.uint8), self._alloc((self._w + 1,) + tuple(depth.shape), np.uint16)))
        for i in range(self._ring):
            self._free.put(i)

    def _put(self, item) -> bool:
        """queue.put that gives up when the consumer has gone away (it stopped iterating early)."""
        while not self._stop:
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    @staticmethod
    def _copy_in(ib, db, k, img, dep, release):
        ib[k] = img
        db[k] = dep
        if release is not None:
            release()

    def _fill(self):
        pool = None
        if self._copy_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=self._copy_threads, thread_name_prefix="dvo-copy")
        try:
            carry = None
            done = False
            while not done and not self._stop:
                t0 = time.perf_counter()
                slot, k, ib, db = None, 0, None, None
                copies = []
                while k < self._w + 1:
                    release = None
                    if carry is not None:
                        img, dep = carry
                        carry = None
                    else:
                        try:
                            item = next(self._frames)
                        except StopIteration:
                            done = True
                            break
                        if isinstance(item, Decoded):
                            img, dep, release = item.img, item.depth, item.release
                        else:
                            img, dep = item
                    if not self._bufs:
                        self._make_buffers(img, dep)
                    if slot is None:
                        slot = self._free.get()
                        if slot is None:
                            if release is not None:
                                release()
                            return
                        ib, db = self._bufs[slot]
                    if img.shape != ib.shape[1:] or dep.shape != db.shape[1:]:
                        raise ValueError("frame {} of the window has shape {} / {}, expected {} / {}".format(
                            k, img.shape, dep.shape, ib.shape[1:], db.shape[1:]))
                    # frames go into the (page-locked) window as they arrive
                    if pool is not None:
                        copies.append(pool.submit(self._copy_in, ib, db, k, img, dep, release))
                    else:
                        self._copy_in(ib, db, k, img, dep, release)
                    k += 1
                for c in copies:
                    c.result()
                if k < 2:
                    break
                carry = (ib[k - 1].copy(), db[k - 1].copy())     # the next window may land in another slot
                self.decode_seconds += time.perf_counter() - t0
                if not self._put((slot, k)):
                    break
        except BaseException as e:                  # surfaced in the consumer thread
            self._err = e
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
            self._put(None)

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray, int]]:
        if self._started:
            raise RuntimeError("FrameWindows can be iterated once")
        self._started = True
        self._thread.start()
        held: List[int] = []
        try:
            while True:
                item = self._q.get()
                if item is None:
                    if self._err is not None:
                        raise self._err
                    return
                slot, n = item
                held.append(slot)
                while len(held) > self._ring - 1:          # slots handed out longer ago than ring - 1 windows are free again
                    self._free.put(held.pop(0))
                ib, db = self._bufs[slot]
                yield ib[:n], db[:n], n
        finally:
            self._stop = True
            self._free.put(None)


def align_windows(aligner, windows: Iterable[Tuple[np.ndarray, np.ndarray, int]], colour: bool, max_distance: float = 5.0):
    """Feeds the windows through `aligner` (BatchAligner / StreamedAligner).  While window k is uploaded and solved the decoder
    thread fills window k+1; inside a window the slices of a StreamedAligner overlap one slice's upload with another's solve.
    -> dict(T (n_pairs,3,4), iters, err, seconds)"""
    T, iters, errs = [], [], []
    t0 = time.perf_counter()

    def keep(res):
        if res is not None:
            T.append(res["T"])
            iters.append(res["iters"])
            errs.append(res["err"])

    if hasattr(aligner, "submit_host"):
        # windows pipelined through the slices (StreamedAligner.submit_host): the GPU does not drain between windows, and the host
        # collects window k's results while window k + 1 is uploading.  A FrameWindows ring of 3 keeps the window in flight valid.
        for img, dep, n in windows:
            keep(aligner.submit_host(img, dep, colour, max_distance))
        keep(aligner.drain())
    else:
        for img, dep, n in windows:
            if colour:
                keep(aligner.align_host_bgr(img, dep, max_distance) if hasattr(aligner, "align_host_bgr") else _align_bgr(aligner, img, dep, max_distance))
            else:
                keep(aligner.align_host(img, dep))
    return {"T": np.concatenate(T) if T else np.zeros((0, 3, 4), np.float32),
            "iters": np.concatenate(iters) if iters else np.zeros(0, np.int32),
            "err": np.concatenate(errs) if errs else np.zeros(0), "seconds": time.perf_counter() - t0}


def align_shard(make_aligner: Callable, source: Callable, n_total: int, rank: int, world: int, window_pairs: int,
                colour: bool, alloc: Callable = None, workers=0, streams: int = 1):
    """BASELINE config 4, one rank's part: rank `rank` of `world` aligns the contiguous range of pairs `shard_pairs` gives it
    (frames lo..hi: one boundary frame is shared with the neighbouring rank).  `source(lo, hi)` yields the frames (or decoder
    thunks) lo..hi-1; `make_aligner(height, width, window_pairs)` builds the aligner once the frame size is known.
    -> dict(T, iters, err, seconds, decode_seconds, lo, hi)"""
    from .batch import shard_pairs
    lo, hi = shard_pairs(n_total - 1, rank, world)
    out = {"T": np.zeros((0, 3, 4), np.float32), "iters": np.zeros(0, np.int32), "err": np.zeros(0), "seconds": 0.0,
           "decode_seconds": 0.0, "lo": lo, "hi": hi}
    if hi <= lo:
        return out
    first = next(iter(source(lo, lo + 1)))
    first = first() if callable(first) else first
    h, w = first[1].shape
    window = max(streams, min(window_pairs, hi - lo))
    aligner = make_aligner(h, w, window)
    try:
        windows = FrameWindows(source(lo, hi + 1), window, alloc=alloc, workers=workers)
        res = align_windows(aligner, windows, colour)
    finally:
        if hasattr(aligner, "close"):
            aligner.close()
    out.update(res)
    out["decode_seconds"] = windows.decode_seconds
    return out


def gather_shards(T_rel: np.ndarray, iters: np.ndarray, n_total: int, device=None):
    """All ranks contribute the pose records of their pair range; every rank gets the sequence's records in pair order."""
    import torch.distributed as dist
    from .batch import shard_pairs, pose_records, allgather_poses
    world = dist.get_world_size()
    counts = [shard_pairs(n_total - 1, r, world)[1] - shard_pairs(n_total - 1, r, world)[0] for r in range(world)]
    rec = allgather_poses(pose_records(T_rel, iters), counts, device=device)
    return rec[:, :12].reshape(-1, 3, 4), rec[:, 12]


def _align_bgr(aligner, bgr, depth, max_distance):
    aligner.ingest_host_bgr(bgr, depth, max_distance)
    aligner.solve_enqueue(bgr.shape[0])
    return aligner.solve_fetch(bgr.shape[0])


# -- report files -------------------------------------------------------------------------------------------------
def _se3_log6(T: np.ndarray) -> List[float]:
    return [float(v) for v in Se3.from_matrix(np.asarray(T, dtype=np.float32)).log().reshape(6)]


def build_report(kind: str, T_rel: np.ndarray, init_pose: Optional[np.ndarray], gt_T_wc: Optional[np.ndarray],
                 timestamps, extra: Optional[dict] = None):
    """-> (report dict, trajectory (n,4,4)).  Keys of the reference's report.json (src/test_dvo.py:323-329):
    "estimated_poses" / "estimated_transforms" = 6-vector logs per frame (the first transform is the identity, :63-65),
    "errors" = Euclidean distance of the estimated to the ground-truth position per frame ("N/A" without ground truth,
    :314-319), plus "type" / "timestamps" as the loaders' additional info; added: the 4x4 forms and ATE / RPE."""
    from . import tum
    T_rel = np.asarray(T_rel, dtype=np.float64)
    traj = compose_trajectory(T_rel, init_pose=(np.eye(4) if init_pose is None else init_pose))
    n = traj.shape[0]
    tf4 = np.tile(np.eye(4), (n, 1, 1))
    tf4[1:, :3, :] = T_rel.reshape(-1, 3, 4)
    report = {"type": kind,
              "estimated_poses": [_se3_log6(P) for P in traj],
              "estimated_transforms": [_se3_log6(P) for P in tf4],
              "errors": (["N/A"] * n if gt_T_wc is None else
                         [float(np.linalg.norm(traj[i, :3, 3] - np.asarray(gt_T_wc)[i, :3, 3])) for i in range(n)]),
              "timestamps": [float(t) for t in np.asarray(timestamps)[:n]],
              "estimated_poses_matrix": traj.reshape(n, 16).tolist(),
              "estimated_transforms_matrix": tf4.reshape(n, 16).tolist()}
    if gt_T_wc is not None:
        report.update(tum.evaluate(traj, np.asarray(gt_T_wc)[:n]))
    if extra:
        report.update(extra)
    return report, traj


def write_report(out_dir: Path, report: dict, traj: np.ndarray) -> None:
    """report.json + report.txt (TUM trajectory format `timestamp tx ty tz qx qy qz qw`, src/test_dvo.py:337-345; the
    reference writes it for TUM data only, here always)."""
    from . import tum
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    with (out_dir / "report.json").open("w") as fp:
        json.dump(report, fp, indent=1)
    tum.write_tum_trajectory(out_dir / "report.txt", report["timestamps"], traj)
