#!/usr/bin/env python3
"""Diagnostic: level-0/1 reduction launch time (host-mode calls with fixed poses) for the build DVO_HIP_LIB selects."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import os
import numpy as np
from dvo_amd import synthetic
from dvo_amd.batch import BatchAligner
B = 256
st = synthetic.make_stream(B + 1, 640, 480, seed=1, device="cuda", as_torch=True)
al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, max_iterations=10)
al.ingest_device(st["gray"].data_ptr(), st["depth"].data_ptr(), B + 1)
T = np.tile(np.eye(4, dtype=np.float32)[:3].reshape(1, 12), (B, 1))
T[:, 3] = 0.004; T[:, 7] = -0.003
prev = np.arange(B, dtype=np.int32)
out = []
for lvl in range(4):
    for _ in range(2):
        al.ctx.normal_equations(lvl, prev, prev + 1, T)
    al.ctx.profile_enable(True); al.ctx.profile_read(True)
    for _ in range(int(os.environ.get('DVO_EXP_LAUNCHES', '10'))):
        al.ctx.normal_equations(lvl, prev, prev + 1, T)
    n, ms = al.ctx.profile_read(True)
    al.ctx.profile_enable(False)
    out.append(1e3 * ms[lvl] / n[lvl])
print("L0 %.1f us  L1 %.1f us  L2 %.1f us  L3 %.1f us" % tuple(out))
