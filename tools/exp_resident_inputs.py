#!/usr/bin/env python3
"""How much of the level-0 reduction launch is waiting for HBM?  256 VGA pairs per launch, (a) every pair on its own frames (the job's case:
records and taps stream from HBM), (b) every pair on the SAME two frames (2.4 MB of records + texels: L2-resident after the first pair).  Same
instructions, same points; the difference is what the memory system costs the launch."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
import dvo_amd
from dvo_amd import synthetic

W, H, B = 640, 480, 256
st = synthetic.make_stream(B + 1, W, H, seed=11, device="cuda")
K = st["K"]
with dvo_amd.HipContext(max_w=W, max_h=H, max_levels=4, n_slots=B + 1, max_pairs=B) as ctx:
    ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
    for k in range(B + 1):
        ctx.upload_frame(k, st["gray"][k], st["depth"][k], 4)
    T = np.tile(np.eye(4, dtype=np.float32)[:3], (B, 1, 1))
    T[:, 0, 3] = 0.002
    for lvl in (0, 1):
        for name, prev, cur in (("own frames ", np.arange(B), np.arange(1, B + 1)), ("same frames", np.zeros(B, int), np.ones(B, int))):
            prev, cur = prev.astype(np.int32), cur.astype(np.int32)
            for _ in range(3):
                ctx.normal_equations(lvl, prev, cur, T)
            ts = []
            for _ in range(20):
                ctx.timer_start()
                ctx.normal_equations(lvl, prev, cur, T)
                ts.append(ctx.timer_stop())
            print("level %d, %s: %.1f us per launch (median of 20, host-mode launch incl. the sum of partials and the fetch)" % (lvl, name, 1e3 * float(np.median(ts))))
