#!/usr/bin/env python3
"""Diagnostic: host-side timeline of StreamedAligner.align_host (page-locked frames)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
from dvo_amd import synthetic, _cabi
from dvo_amd.batch import StreamedAligner

B = 512
st = synthetic.make_stream(B + 1, 640, 480, seed=1, device="cuda")
gp = _cabi.pinned_array(st["gray"].shape, np.uint8); gp[...] = st["gray"]
dp = _cabi.pinned_array(st["depth"].shape, np.uint16); dp[...] = st["depth"]
for streams in (2, 4):
    sa = StreamedAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, n_streams=streams, max_iterations=10)
    sa.align_host(gp, dp)
    for rep in range(3):
        sl = sa._slices(B + 1)
        marks = [("start", time.perf_counter())]
        for a, lo, hi in sl:
            a.ingest_host(gp[lo:hi + 1], dp[lo:hi + 1])
            marks.append(("ingest%d" % lo, time.perf_counter()))
            a.solve_enqueue(hi - lo + 1)
            marks.append(("enq%d" % lo, time.perf_counter()))
        for a, lo, hi in sl:
            a.solve_fetch(hi - lo + 1)
            marks.append(("fetch%d" % lo, time.perf_counter()))
        t0 = marks[0][1]
        print("streams=%d: " % streams + "  ".join("%s %.2f" % (n, 1e3 * (t - t0)) for n, t in marks[1:]))
    sa.close()
