import os, sys, numpy as np
sys.path.insert(0, "dense-visual-odometry_amd"); sys.path.insert(0, "tests")
from dvo_amd.batch import BatchAligner
from dvo_amd import synthetic
d = np.load("tests/golden/frames.npz") if os.path.exists("tests/golden/frames.npz") else None
print("fixture keys", None if d is None else list(d.keys())[:12])
st = synthetic.make_stream(33, 640, 480, seed=1)
cases = [("synthetic", st["gray"], st["depth"], st["K"], st["depth_scale"])]
for name, gray, depth, K, scale in cases:
    out = []
    for exact in (False, True):
        os.environ.pop("DVO_ST_EXACT", None)
        if exact: os.environ["DVO_ST_EXACT"] = "1"
        al = BatchAligner(K, scale, 640, 480, 4, max_pairs=32, max_iterations=10, weighting="student_t_reference")
        out.append(al.align_host(gray, depth))
        print(name, "exact" if exact else "one-pass", "fallbacks", al.ctx.student_t_fallbacks(), "small kept/redone", al.ctx.student_t_small_level_counts())
        al.close()
    os.environ.pop("DVO_ST_EXACT", None)
    print(name, "iters equal", np.array_equal(out[0]["iters"], out[1]["iters"]), "max|dxi|", np.max(np.abs(out[0]["xi"] - out[1]["xi"])),
          "max rel derr", np.max(np.abs(out[0]["err"]/out[1]["err"] - 1)))
