#!/usr/bin/env python3
"""Where the sequence front end binds (the reference harness decodes every frame into RAM before its first step,
src/test_dvo.py:180-191, and times step() on decoded frames, :305-309): a TUM-layout dataset of PNGs on disk, read by
dvo_amd.sequence.FrameWindows (decoder thread pool -> page-locked windows) and aligned by the batched device solve.

    python tools/exp_sequence_png.py [--frames 1200] [--distinct 60] [--threads 4,8,16,32] [--processes 8,16,32] [--window 128]

Reports, per decoder-thread count and per decoder-PROCESS count (dvo_amd.sequence.DecoderPool): decode-only frames/s (no GPU work),
end-to-end frame-pairs/s (decode + upload + colour ingest + solve), and beside them the same windows fed from memory (the GPU-bound
rate of this front end).  `measure()` is also the
`sequence_png` leg of bench.py (bounded to a few seconds)."""
import argparse
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np  # noqa: E402


def measure(n_frames=480, n_distinct=48, threads=(8, 16), window=128, streams=2, device=0, colour=True, levels=4, iterations=10,
            decode_only=True, from_memory=True, seed=77, log=None, pools=()):
    """pools: DecoderPool objects (worker PROCESSES; created by the caller before the GPU was touched) measured like the thread counts."""
    from dvo_amd import synthetic, tum, sequence, _cabi
    from dvo_amd.batch import StreamedAligner
    W, H = 640, 480
    st = synthetic.make_stream(n_distinct, W, H, seed=seed, device="cuda")
    out = {"frames": n_frames, "distinct_png_files": n_distinct, "image": "%dx%d, 8-bit RGB PNG + 16-bit depth PNG (PIL, default compression)" % (W, H),
           "window_pairs": window, "streams": streams, "input": "colour (BGR2GRAY + depth clamp on the device)" if colour else "gray"}
    try:
        out["usable_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        out["usable_cpus"] = os.cpu_count()
    with tempfile.TemporaryDirectory(prefix="dvo_png_") as tmp:
        t0 = time.perf_counter()
        tum.write_tum_dataset(tmp, st["gray"], st["depth"], st["T_wc"], n_listed=n_frames, K=st["K"], depth_scale=st["depth_scale"])
        out["dataset_write_s"] = time.perf_counter() - t0
        seq = tum.load_tum_sequence(tmp)
        rgb_bytes = sum(os.path.getsize(p) for p in set(seq.rgb_paths)) / len(set(seq.rgb_paths))
        dep_bytes = sum(os.path.getsize(p) for p in set(seq.depth_paths)) / len(set(seq.depth_paths))
        out["png_bytes_per_frame"] = {"rgb": rgb_bytes, "depth": dep_bytes}
        n = len(seq)

        def thunks():
            import functools
            for k in range(n):
                yield functools.partial(_read if colour else _read_gray, seq.rgb_paths[k], seq.depth_paths[k])

        al = StreamedAligner(st["K"], st["depth_scale"], W, H, levels, max_pairs=window, n_streams=streams, device=device, max_iterations=iterations)
        try:
            # warm-up: graphs captured, page cache warm
            # one ring of page-locked windows for every measurement below (page-locking 3 x 129 VGA colour frames takes ~0.1 s: a start-up
            # cost of a run, not part of its rate)
            warm = sequence.FrameWindows(list(thunks())[:window + 1], window, alloc=_cabi.pinned_array, workers=max(threads))
            sequence.align_windows(al, warm, colour)
            ring = warm.buffers
            rows = []
            for w in list(threads) + list(pools):
                is_pool = not isinstance(w, int)
                row = {"decoder_processes": len(w._procs)} if is_pool else {"decoder_threads": w}
                if decode_only:
                    t0 = time.perf_counter()
                    fw = sequence.FrameWindows(thunks(), window, workers=w, buffers=ring)
                    cnt = sum(k - 1 for _, _, k in fw) + 1
                    row["decode_only_frames_per_s"] = cnt / (time.perf_counter() - t0)
                fw = sequence.FrameWindows(thunks(), window, workers=w, buffers=ring)
                res = sequence.align_windows(al, fw, colour)
                row["frame_pairs_per_s"] = res["T"].shape[0] / res["seconds"]
                row["decoder_thread_busy_s"] = fw.decode_seconds
                rows.append(row)
                if log:
                    log("  %2d %s: decode only %7.0f frames/s   decode + upload + solve %7.0f frame-pairs/s" % (
                        len(w._procs) if is_pool else w, "processes" if is_pool else "threads  ", row.get("decode_only_frames_per_s", float("nan")),
                        row["frame_pairs_per_s"]))
            out["by_decoder_threads"] = rows
            if from_memory:
                frames = [(_read if colour else _read_gray)(seq.rgb_paths[k], seq.depth_paths[k]) for k in range(min(n, n_distinct))]
                period = max(2 * (len(frames) - 1), 1)
                idx = [(i % period) if (i % period) < len(frames) else period - (i % period) for i in range(n)]
                fw = sequence.FrameWindows((frames[i] for i in idx), window, buffers=ring)
                res = sequence.align_windows(al, fw, colour)
                out["from_memory_frame_pairs_per_s"] = res["T"].shape[0] / res["seconds"]
                out["from_memory_fill_thread_busy_s"] = fw.decode_seconds
                if log:
                    log("  decoded frames in memory (memcpy into the windows): %7.0f frame-pairs/s  (fill thread busy %.0f of %.0f ms)" % (
                        out["from_memory_frame_pairs_per_s"], 1e3 * fw.decode_seconds, 1e3 * res["seconds"]))
        finally:
            al.close()
    best = max(out["by_decoder_threads"], key=lambda r: r["frame_pairs_per_s"])
    out["best"] = {k: v for k, v in best.items() if k in ("decoder_threads", "decoder_processes", "frame_pairs_per_s")}
    return out


def _read(rgb_path, depth_path):
    from dvo_amd import tum
    return tum.read_frame(rgb_path, depth_path, "bgr")


def _read_gray(rgb_path, depth_path):
    c, d = _read(rgb_path, depth_path)
    return np.ascontiguousarray(c[..., 1]), d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1200)
    ap.add_argument("--distinct", type=int, default=60)
    ap.add_argument("--threads", default="4,8,16,32")
    ap.add_argument("--processes", default="8,16,32", help="decoder PROCESS counts (DecoderPool), beside the thread counts")
    ap.add_argument("--window", type=int, default=128)
    ap.add_argument("--streams", type=int, default=2)
    args = ap.parse_args()
    threads = tuple(int(v) for v in args.threads.split(","))
    procs = tuple(int(v) for v in args.processes.split(",")) if args.processes else ()
    from dvo_amd import sequence
    # the worker processes exist before this process touches the GPU (measure() renders its frames on it)
    pools_c = [sequence.DecoderPool(p, (480, 640, 3), (480, 640)) for p in procs]
    pools_g = [sequence.DecoderPool(p, (480, 640), (480, 640)) for p in procs[-1:]]
    for colour in (True, False):
        print("== %s input, %d frames listed (%d distinct PNG pairs), windows of %d pairs in %d streams" % (
            "colour" if colour else "gray", args.frames, args.distinct, args.window, args.streams))
        out = measure(args.frames, args.distinct, threads, args.window, args.streams, colour=colour, log=print, pools=(pools_c if colour else pools_g))
        print("  PNG bytes per frame: rgb %.0f, depth %.0f; usable CPUs %s; dataset written in %.1f s" % (
            out["png_bytes_per_frame"]["rgb"], out["png_bytes_per_frame"]["depth"], out["usable_cpus"], out["dataset_write_s"]))
    for p in pools_c + pools_g:
        p.close()


if __name__ == "__main__":
    main()
