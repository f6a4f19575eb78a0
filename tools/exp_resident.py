#!/usr/bin/env python3
"""How much of the level-0 reduction's time is the memory system?  The same launch (256 pairs, VGA) with every pair reading the SAME two
frame slots (2.9 MB working set: everything comes from the L2) against 256 different slot pairs (756 MB per launch from HBM)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
import dvo_amd
from dvo_amd import synthetic, _cabi

B = 256
st = synthetic.make_stream(B + 1, 640, 480, seed=0, device="cuda")
ctx = _cabi.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=B + 1, max_pairs=B)
K = np.asarray(st["K"], dtype=np.float32)
ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
ctx.upload_frames(np.arange(B + 1, dtype=np.int32), st["gray"], st["depth"], 4)
for _ in range(3):                                                           # warm-up: clocks, graph capture
    ctx.solve_batch(np.arange(B, dtype=np.int32), np.arange(B, dtype=np.int32) + 1, levels=4, max_iterations=10)
for name, prev, cur in (("256 different slot pairs", np.arange(B), np.arange(B) + 1), ("one slot pair 256 times", np.zeros(B, int), np.ones(B, int)),
                        ("8 slot pairs", np.arange(B) % 8, np.arange(B) % 8 + 1), ("32 slot pairs (94 MB)", np.arange(B) % 32, np.arange(B) % 32 + 1),
                        ("64 slot pairs (189 MB)", np.arange(B) % 64, np.arange(B) % 64 + 1), ("128 slot pairs (378 MB)", np.arange(B) % 128, np.arange(B) % 128 + 1),
                        ("256 different slot pairs", np.arange(B), np.arange(B) + 1)):
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    for _ in range(4):
        ctx.solve_batch(prev.astype(np.int32), cur.astype(np.int32), levels=4, max_iterations=10)
    launches, ms = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    print("%-26s level 0: %.1f us per launch (%d launches)   level 1: %.1f us" % (name, 1e3 * ms[0] / launches[0], launches[0], 1e3 * ms[1] / max(launches[1], 1)))
