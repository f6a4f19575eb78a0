#!/usr/bin/env python3
"""Per-frame latency of the drop-in step() API (one frame pair at a time) on the GPU box."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
import dvo_amd
from dvo_amd import synthetic

st = synthetic.make_stream(12, 640, 480, seed=3, device="cuda")
cam = dvo_amd.RGBDCameraModel(st["K"], st["depth_scale"])
for solver in ("host", "device"):
    dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=10, solver=solver)
    color = [np.repeat(g[..., None], 3, axis=2) for g in st["gray"]]
    ts = []
    for k in range(12):
        d = st["depth"][k].copy()
        t0 = time.perf_counter()
        dvo.step(color[k], d)
        ts.append(time.perf_counter() - t0)
    print("solver=%s: step() ms per frame (frames 2..12): median %.2f min %.2f max %.2f ; iterations %d" % (
        solver, 1e3 * np.median(ts[1:]), 1e3 * min(ts[1:]), 1e3 * max(ts[1:]), dvo.last_iterations))
    # gray path without colour conversion / depth mutation
    dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=10, solver=solver)
    ts = []
    for k in range(12):
        t0 = time.perf_counter()
        dvo.step_gray(st["gray"][k], st["depth"][k])
        ts.append(time.perf_counter() - t0)
    print("solver=%s: step_gray() ms per frame: median %.2f min %.2f" % (solver, 1e3 * np.median(ts[1:]), 1e3 * min(ts[1:])))
# Student-t reference weighter (use_weighter=True), device solve
dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=10, solver="device", use_weighter=True)
ts = []
for k in range(12):
    t0 = time.perf_counter()
    dvo.step_gray(st["gray"][k], st["depth"][k])
    ts.append(time.perf_counter() - t0)
print("solver=device, use_weighter=True: step_gray() ms per frame: median %.2f min %.2f" % (1e3 * np.median(ts[1:]), 1e3 * min(ts[1:])))
# the build's true Gauss-Newton mode, 3 iterations per level
dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=3, solver="device", reference_variant="gn")
ts = []
for k in range(12):
    t0 = time.perf_counter()
    dvo.step_gray(st["gray"][k], st["depth"][k])
    ts.append(time.perf_counter() - t0)
print("solver=device, reference_variant='gn', 3 iterations per level: step_gray() ms per frame: median %.2f min %.2f" % (1e3 * np.median(ts[1:]), 1e3 * min(ts[1:])))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(24): dvo.step_gray(st["gray"][k % 12], st["depth"][k % 12])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
