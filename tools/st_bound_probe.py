#!/usr/bin/env python3
"""Distribution of the one-pass Student-t bound (trace slots 30 = bound / |inc|, 31 = eps) per level and form, with the keep
threshold out of the way (experiment build, DVO_ST_TOL=1): fixture pairs and a rendered batch.  python tools/st_bound_probe.py"""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
os.environ.setdefault("DVO_HIP_LIB", str(ROOT / "ab_libs" / "libdvo_hip_exp.so"))
os.environ["DVO_ST_TOL"] = "1"
import dvo_amd
from dvo_amd import _cabi, synthetic
import dvo_oracle as O

f = np.load(ROOT / "tests" / "golden" / "frames.npz")
sets = [("fixture", f["gray"], f["depth"], f["intrinsics"], float(f["depth_scale"]))]
st = synthetic.make_stream(17, 640, 480, seed=1000)
sets.append(("rendered", st["gray"], st["depth"], st["K"], st["depth_scale"]))
for name, gray, depth, K, sc in sets:
    n = gray.shape[0] - 1
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=n + 1, max_pairs=n) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], sc)
        for k in range(n + 1):
            ctx.upload_frame(k, gray[k], depth[k], 4)
        ctx.trace_enable(40)
        prev = np.arange(n, dtype=np.int32)
        res = ctx.solve_batch(prev, prev + 1, levels=4, max_iterations=10, weight_mode=_cabi.WEIGHT_STUDENT_T)
        tr = ctx.trace_read(n)
        print("%s: %d pairs, fallback pairs %d" % (name, n, ctx.student_t_fallback_pairs()))
        # iterations 0-9 level 3, 10-19 level 2, 20-29 level 1, 30-39 level 0
        for lvl, lo in ((2, 10), (1, 20), (0, 30)):
            for form, its in (("moment/exact", range(lo, lo + 3)), ("predicted", range(lo + 3, lo + 10))):
                b = tr[:, list(its), 30].ravel()
                e = tr[:, list(its), 31].ravel()
                b = b[b > 0]
                if b.size:
                    print("  level %d %-12s bound/|inc|: median %.2e  p90 %.2e  max %.2e   eps: median %.2e max %.2e" % (
                        lvl, form, np.median(b), np.percentile(b, 90), b.max(), np.median(e[e > 0]) if (e > 0).any() else 0, e.max()))
        # true error of the one-pass increments, teacher-forced, for pair 0
        gp0, dp0, gp1 = O.build_pyramid(gray[0], 4), O.build_pyramid(depth[0], 4), O.build_pyramid(gray[1], 4)
        worst = {}
        for j in range(20, 40):
            lvl = 1 if j < 30 else 0
            T = np.eye(4, dtype=np.float32)
            T[:3] = tr[0, j, 38:50].reshape(3, 4).astype(np.float32)
            lv = O.PairLevel(gp0[lvl], dp0[lvl], gp1[lvl], O.level_intrinsics(K, lvl), sc)
            r, J, _, _ = O.residuals_and_jacobian(lv, T)
            w, _ = O.student_t_weights(r)
            H, b, _, _ = O.normal_equations(r, J, w)
            inc = O.solve_increment(H, b)
            rel = np.linalg.norm(tr[0, j, 32:38] - inc) / np.linalg.norm(inc)
            print("    pair 0 it %2d level %d: true rel err %.2e   bound %.2e  eps %.2e" % (j, lvl, rel, tr[0, j, 30], tr[0, j, 31]))
