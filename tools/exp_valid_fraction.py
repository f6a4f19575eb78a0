#!/usr/bin/env python3
"""Experiment: level-0 reduction time vs depth-valid fraction (how much do depth-less lanes cost?)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np, torch
from dvo_amd import synthetic
from dvo_amd.batch import BatchAligner

B = 256
for frac in (0.2, 0.0, 0.5):
    st = synthetic.make_stream(B + 1, 640, 480, seed=1, invalid_fraction=frac, device="cuda", as_torch=True)
    d = st["depth"]
    if frac == 0.0:
        d = d.clone(); d[d == 0] = 12000          # fill the border too
    al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, max_iterations=10)
    g = st["gray"]
    for _ in range(2):
        al.align_device(g.data_ptr(), d.data_ptr(), B + 1)
    al.ctx.profile_enable(True); al.ctx.profile_read(True)
    for _ in range(3):
        al.align_device(g.data_ptr(), d.data_ptr(), B + 1)
    n, ms = al.ctx.profile_read(True)
    valid = float((d[:-1] != 0).float().mean())
    print("invalid_fraction=%.1f valid=%.3f  L0 %.1f us  L1 %.1f us" % (frac, valid, 1e3 * ms[0] / n[0], 1e3 * ms[1] / n[1]))
    al.close()
