#!/usr/bin/env python3
"""Does the job gain from keeping the streams of one GPU out of phase?  Four sub-batches per step in four streams:
  lockstep   every step enqueues all four slices, then fetches all four (the streams run ingest / small levels / level 1 / level 0 in phase);
  pipelined  slice by slice: fetch the slice's previous result, enqueue its next window (no global drain between steps);
  staggered  pipelined, the first windows started a quarter of a step apart (host sleep), so that one stream's memory-bound ingest
             and latency-bound small levels run beside the other streams' level-0 reductions.
python tools/exp_pipeline.py [--batch 1024] [--steps 40] [--weighting none]"""
import argparse
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--weighting", default="none")
args = ap.parse_args()
from dvo_amd import synthetic
from dvo_amd.batch import StreamedAligner
B, S, K = args.batch, args.streams, args.steps
W, H, L = 640, 480, 4
st = synthetic.make_stream(B + 1, W, H, seed=1, device="cuda", as_torch=True)
gray, depth = st["gray"], st["depth"]
al = StreamedAligner(st["K"], st["depth_scale"], W, H, L, max_pairs=B, n_streams=S, max_iterations=10, weighting=args.weighting)
npx = W * H
sl = al._slices(B + 1)


def sync():
    al.synchronize()
    torch.cuda.synchronize()


def enqueue(a, lo, hi):
    a.ingest_device(gray.data_ptr() + lo * npx, depth.data_ptr() + 2 * lo * npx, hi - lo + 1)
    a.solve_enqueue(hi - lo + 1)


def lockstep():
    for _ in range(K):
        al.align_device(gray.data_ptr(), depth.data_ptr(), B + 1)


def pipelined(stagger_s=0.0):
    pending = [False] * len(sl)
    for k in range(K):
        for i, (a, lo, hi) in enumerate(sl):
            if pending[i]:
                a.solve_fetch(hi - lo + 1)
            elif stagger_s > 0.0 and i > 0:
                time.sleep(stagger_s)
            enqueue(a, lo, hi)
            pending[i] = True
    for i, (a, lo, hi) in enumerate(sl):
        a.solve_fetch(hi - lo + 1)


for _ in range(3):
    al.align_device(gray.data_ptr(), depth.data_ptr(), B + 1)
sync()
for rnd in range(3):
    out = []
    for name, fn in (("lockstep", lockstep), ("pipelined", pipelined), ("staggered", None)):
        sync()
        t0 = time.perf_counter()
        if fn is None:
            pipelined(stagger_s=step_s / S)
        else:
            fn()
        sync()
        dt = time.perf_counter() - t0
        if name == "lockstep":
            step_s = dt / K
        out.append("%s %.1f k pairs/s (%.3f ms/step)" % (name, 1e-3 * B * K / dt, 1e3 * dt / K))
    print("round %d: " % rnd + "   ".join(out), flush=True)
al.close()
