#!/usr/bin/env python3
"""Where the persistent small-level kernel (k_gn_small) spends its time: isolated launch time per workgroup size and the in-kernel
phase clocks of thread 0 (DVO_SMALL_STAMPS).   python tools/exp_small.py [--pairs 256] [--weighting none]"""
import argparse
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np

# needs the experiment build of the library: make -C dense-visual-odometry_amd/csrc experiments
os.environ.setdefault("DVO_HIP_LIB", str(ROOT / "ab_libs" / "libdvo_hip_exp.so"))

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=256)
ap.add_argument("--weighting", default="none")
ap.add_argument("--threads", default="256,512,1024")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
from dvo_amd import synthetic
from dvo_amd.batch import BatchAligner
B = args.pairs
st = synthetic.make_stream(B + 1, 640, 480, seed=1, device="cuda", as_torch=True)
os.environ["DVO_SMALL_STAMPS"] = "1"
names = ["record loops", "wave reduce", "cross-wave sum", "control", "t-scale passes", "kernel"]
for thr in [int(t) for t in args.threads.split(",")]:
    os.environ["DVO_SMALL_THREADS"] = str(thr)
    al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, max_iterations=10, weighting=args.weighting)
    al.ingest_device(st["gray"].data_ptr(), st["depth"].data_ptr(), B + 1)
    al.solve_enqueue(B + 1); al.solve_fetch(B + 1)
    al.ctx.small_stamps(B, reset=True)
    al.ctx.profile_enable(True); al.ctx.profile_read(True)
    for _ in range(args.reps):
        al.solve_enqueue(B + 1); res = al.solve_fetch(B + 1)
    n, ms = al.ctx.profile_read(True)
    al.ctx.profile_enable(False)
    s = al.ctx.small_stamps(B, reset=True).astype(np.float64) / args.reps
    tot = s[:, 5].mean()
    print("threads %4d: k_gn_small %.1f us per launch of %d pairs (events); thread-0 clocks per launch, mean over pairs: total %.0f kcycles; " % (
        thr, 1e3 * ms[3] / max(n[3], 1), B, tot / 1e3) + "  ".join("%s %.1f %%" % (names[k], 100 * s[:, k].mean() / tot) for k in range(5)))
    al.close()
