#!/usr/bin/env python3
"""Where a window of host frames spends its time: copy into a page-locked window (1 / 4 threads), upload + ingest + solve of the window."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from dvo_amd import _cabi, synthetic
from dvo_amd.batch import StreamedAligner

W, H, n = 640, 480, 129
st = synthetic.make_stream(8, W, H, seed=3, device="cuda")
src = [(np.repeat(st["gray"][k % 8][..., None], 3, axis=2).copy(), st["depth"][k % 8].copy()) for k in range(n)]
for name, alloc in (("pageable", lambda s, d: np.empty(s, d)), ("page-locked", _cabi.pinned_array)):
    ib, db = alloc((n, H, W, 3), np.uint8), alloc((n, H, W), np.uint16)

    def one(k):
        ib[k] = src[k][0]
        db[k] = src[k][1]
    for rep in range(2):
        t = time.perf_counter()
        for k in range(n):
            one(k)
        dt = time.perf_counter() - t
    print("%-12s window of %d colour frames: 1 thread %.2f ms (%.0f frames/s)" % (name, n, dt * 1e3, n / dt))
    for w in (4, 8):
        with ThreadPoolExecutor(w) as ex:
            list(ex.map(one, range(n)))
            t = time.perf_counter()
            list(ex.map(one, range(n)))
            dt = time.perf_counter() - t
        print("%-12s %d threads %.2f ms (%.0f frames/s)" % ("", w, dt * 1e3, n / dt))
al = StreamedAligner(st["K"], st["depth_scale"], W, H, 4, max_pairs=128, n_streams=2, max_iterations=10)
al.align_host_bgr(ib, db)
for rep in range(3):
    t = time.perf_counter()
    al.align_host_bgr(ib, db)
    dt = time.perf_counter() - t
    print("align_host_bgr of the page-locked window: %.2f ms (%.0f pairs/s)" % (dt * 1e3, (n - 1) / dt))
t = time.perf_counter()
for rep in range(6):
    al.submit_host(ib, db, True)
al.drain()
dt = (time.perf_counter() - t) / 6
print("submit_host x 6 (same window): %.2f ms per window (%.0f pairs/s)" % (dt * 1e3, (n - 1) / dt))
al.close()

# the same through FrameWindows (decoded frames in memory -> ring of page-locked windows -> submit_host), with the time the consumer waits
# for a window beside the time it spends submitting it
from dvo_amd import sequence
al = StreamedAligner(st["K"], st["depth_scale"], W, H, 4, max_pairs=128, n_streams=2, max_iterations=10)
ring = None
for colour in (True, False):
    frames = [(s[0] if colour else np.ascontiguousarray(s[0][..., 1]), s[1]) for s in src]
    for threads in (1, 4, 8):
        ring = None
        for rep in range(2):
            fw = sequence.FrameWindows((frames[i % n] for i in range(1200)), 128, alloc=_cabi.pinned_array, copy_threads=threads, buffers=ring)
            t_wait = t_sub = 0.0
            t0 = time.perf_counter()
            it = iter(fw)
            pairs = 0
            while True:
                ta = time.perf_counter()
                try:
                    img, dep, k = next(it)
                except StopIteration:
                    break
                tb = time.perf_counter()
                al.submit_host(img, dep, colour)
                tc = time.perf_counter()
                t_wait += tb - ta
                t_sub += tc - tb
                pairs += k - 1
            al.drain()
            dt = time.perf_counter() - t0
            ring = fw.buffers
        print("%s, %d copy threads: %.0f pairs/s; waiting for windows %.1f ms, submitting %.1f ms, fill thread busy %.1f ms (of %.1f)" % (
            "colour" if colour else "gray", threads, pairs / dt, t_wait * 1e3, t_sub * 1e3, fw.decode_seconds * 1e3, dt * 1e3))
al.close()
