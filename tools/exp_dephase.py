#!/usr/bin/env python3
"""Experiment: two solve chains in lockstep vs offset by half a step (do the coarse levels of one hide under the
level-0 work of the other?)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np, torch
from dvo_amd import synthetic
from dvo_amd.batch import BatchAligner

B = 256
st = synthetic.make_stream(2 * B + 1, 640, 480, seed=1, device="cuda", as_torch=True)
g, d = st["gray"], st["depth"]
npx = 640 * 480
parts = [BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, max_iterations=10) for _ in range(2)]
ptrs = [(g.data_ptr() + i * B * npx, d.data_ptr() + 2 * i * B * npx) for i in range(2)]

def enqueue(i):
    parts[i].ingest_device(ptrs[i][0], ptrs[i][1], B + 1)
    parts[i].solve_enqueue(B + 1)

def run(offset_s, steps=20):
    for i in range(2):
        enqueue(i); parts[i].solve_fetch(B + 1)
    enqueue(0)
    if offset_s:
        time.sleep(offset_s)
    enqueue(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        parts[0].solve_fetch(B + 1); enqueue(0)
        parts[1].solve_fetch(B + 1); enqueue(1)
    parts[0].solve_fetch(B + 1); parts[1].solve_fetch(B + 1)
    dt = time.perf_counter() - t0
    return 2 * B * (steps + 1) / dt          # the two enqueued before t0 finish inside the window; count steps+1 minus...

for off in (0.0, 0.002, 0.004, 0.0, 0.002, 0.004):
    print("offset %.1f ms: %.0f frame-pairs/s" % (off * 1e3, run(off)))
