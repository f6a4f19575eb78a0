#!/usr/bin/env python3
"""A/B of the tap formulations of the level-0 reduction (dvo_exp_taps.inc) inside one process: isolated launch time of 256 VGA
pairs (HIP events, fixed poses, host-mode calls) for the product kernel (mode 0) and the experimental kernels (DVO_EXP_TAPS=1..5),
interleaved rounds, and a bitwise comparison of the normal-equation records with the product kernel's.
    python tools/exp_taps.py [--modes 0,1,2,3,4] [--rounds 3] [--launches 40] [--levels 0,1] [--width 640 --height 480] [--rot 0.5]"""
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
ap.add_argument("--modes", default="0,1,2,3,4,5")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--levels", default="0,1")
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--pairs", type=int, default=256)
ap.add_argument("--rot", type=float, default=0.5, help="degrees per frame of the rendered stream")
ap.add_argument("--pose", default="near", choices=["near", "identity"], help="poses of the timed launches: the stream's own motion (converged) or identity (first iteration)")
args = ap.parse_args()
from dvo_amd import synthetic
from dvo_amd.batch import BatchAligner

B = args.pairs
modes = [int(m) for m in args.modes.split(",")]
levels = [int(l) for l in args.levels.split(",")]
st = synthetic.make_stream(B + 1, args.width, args.height, seed=1, step_rot_deg=args.rot, device="cuda", as_torch=True)
nlev = 4
T = np.tile(np.eye(4, dtype=np.float32)[:3].reshape(1, 12), (B, 1))
if args.pose == "near":
    for i in range(B):
        T[i] = synthetic.relative_transform(st["T_wc"][i], st["T_wc"][i + 1])[:3].reshape(12).astype(np.float32)
prev = np.arange(B, dtype=np.int32)
als = {}
for m in modes:
    if m:
        os.environ["DVO_EXP_TAPS"] = str(m)
    else:
        os.environ.pop("DVO_EXP_TAPS", None)
    al = BatchAligner(st["K"], st["depth_scale"], args.width, args.height, nlev, max_pairs=B, max_iterations=10)
    al.ingest_device(st["gray"].data_ptr(), st["depth"].data_ptr(), B + 1)
    als[m] = al
os.environ.pop("DVO_EXP_TAPS", None)
ref = {}
for lvl in levels:
    for m in modes:
        out = als[m].ctx.normal_equations(lvl, prev, prev + 1, T)
        if m == 0:
            ref[lvl] = out.copy()
        elif 0 in modes:
            same = np.array_equal(out, ref[lvl])
            err = float(np.max(np.abs(out - ref[lvl]) / (np.abs(ref[lvl]) + 1e-30)))
            print("level %d mode %d: records %s (max rel diff %.3g), N sum %d" % (lvl, m, "bit-identical" if same else "DIFFER", err, int(out[:, 28].sum())))
for rnd in range(args.rounds):
    for lvl in levels:
        line = []
        for m in modes:
            ctx = als[m].ctx
            for _ in range(2):
                ctx.normal_equations(lvl, prev, prev + 1, T)
            ctx.profile_enable(True); ctx.profile_read(True)
            for _ in range(args.launches):
                ctx.normal_equations(lvl, prev, prev + 1, T)
            n, ms = ctx.profile_read(True)
            ctx.profile_enable(False)
            line.append("mode %d %.1f us" % (m, 1e3 * ms[lvl] / n[lvl]))
        print("round %d level %d (%dx%d, %d pairs, rot %.2f, pose %s): %s" % (rnd, lvl, args.width >> lvl, args.height >> lvl, B, args.rot, args.pose, "  ".join(line)))
