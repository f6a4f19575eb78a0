#!/usr/bin/env python3
"""Cost of the per-pair Student-t fallback: a batch of 256 VGA pairs of the reference's robust mode with NO / ONE /
EIGHT pairs that trip the one-pass bound (identical frames + 40 inverted pixels), (a) alone on one stream and (b) as one of four
slices of the 1024-pair job.   python tools/exp_st_fallback.py [--reps 10]"""
import argparse
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--pairs", type=int, default=256)
args = ap.parse_args()
from dvo_amd import synthetic
from dvo_amd.batch import BatchAligner, StreamedAligner

B, W, H, L = args.pairs, 640, 480, 4
st = synthetic.make_stream(4 * B + 1, W, H, seed=1000, device="cuda", as_torch=True)


def with_trips(gray, depth, where):
    """frames (k, k+1) made identical except 40 inverted pixels for every k in `where`"""
    g, d = gray.clone(), depth.clone()
    rng = np.random.default_rng(3)
    for k in where:
        g[k + 1] = g[k]
        d[k + 1] = d[k]
        ys = torch.as_tensor(rng.integers(40, 440, 40), device="cuda")
        xs = torch.as_tensor(rng.integers(40, 600, 40), device="cuda")
        g[k + 1, ys, xs] = 255 - g[k + 1, ys, xs]
    return g.contiguous(), d.contiguous()


def timed(fn, sync, reps):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


print("(a) one stream, %d pairs per batch" % B)
al = BatchAligner(st["K"], st["depth_scale"], W, H, L, max_pairs=B, max_iterations=10, weighting="student_t_reference")
base = None
for name, where in (("clean", []), ("1 tripping pair", [B // 2]), ("8 tripping pairs", list(range(10, B, B // 8))[:8])):
    g, d = with_trips(st["gray"][:B + 1], st["depth"][:B + 1], where)
    n0 = al.ctx.student_t_fallback_pairs()
    dt = timed(lambda: al.align_device(g.data_ptr(), d.data_ptr(), B + 1), lambda: (al.ctx.synchronize(), torch.cuda.synchronize()), args.reps)
    per = (al.ctx.student_t_fallback_pairs() - n0) / (args.reps + 1)
    base = base or dt
    print("  %-18s %7.3f ms per batch  (%.3f x clean)   pairs solved again per batch: %.1f" % (name, 1e3 * dt, dt / base, per))
al.close()

print("(b) four slices of a %d-pair window (the bench's job), pipelined windows" % (4 * B))
sa = StreamedAligner(st["K"], st["depth_scale"], W, H, L, max_pairs=4 * B, n_streams=4, max_iterations=10, weighting="student_t_reference")
base = None
for name, where in (("clean", []), ("1 tripping pair per slice", [B // 2 + s * B for s in range(4)]),
                    ("8 tripping pairs per slice", [k + s * B for s in range(4) for k in list(range(10, B - 1, B // 8))[:8]])):
    g, d = with_trips(st["gray"], st["depth"], where)

    def run():
        for _ in range(3):
            sa.submit_device(g.data_ptr(), d.data_ptr(), 4 * B + 1)
        sa.drain()
    n0 = sum(a.ctx.student_t_fallback_pairs() for a in sa.parts)
    dt = timed(run, lambda: (sa.synchronize(), torch.cuda.synchronize()), max(2, args.reps // 3)) / 3
    per = (sum(a.ctx.student_t_fallback_pairs() for a in sa.parts) - n0) / (3 * (max(2, args.reps // 3) + 1))
    base = base or dt
    print("  %-28s %7.3f ms per window = %6.0f pairs/s  (%.3f x clean)   pairs solved again per window: %.1f" % (name, 1e3 * dt, 4 * B / dt, dt / base, per))
sa.close()
