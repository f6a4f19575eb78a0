#!/usr/bin/env python3
"""Single-pair step_gray() loop on the device solver (for kernel traces): python tools/latency_one.py [n_frames]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
import dvo_amd
from dvo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
st = synthetic.make_stream(12, 640, 480, seed=3, device="cuda")
cam = dvo_amd.RGBDCameraModel(st["K"], st["depth_scale"])
dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=10, solver="device")
ts = []
for k in range(n):
    t0 = time.perf_counter()
    dvo.step_gray(st["gray"][k % 12], st["depth"][k % 12])
    ts.append(time.perf_counter() - t0)
print("step_gray ms per frame: median %.3f min %.3f" % (1e3 * np.median(ts[1:]), 1e3 * min(ts[1:])))
