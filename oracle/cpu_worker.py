"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): worker of bench.py's `cpu_baseline` leg.  Runs the NumPy
oracle (float32 J^T J like the reference, base_robust_dvo.py:168-188) on a range of frame pairs, one BLAS thread; imported
by spawned worker processes, so it must not import torch or touch the GPU."""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dvo_oracle as O


def run_pairs(path, lo, hi, levels, iterations):
    """-> (seconds, pairs) for pairs lo..hi-1 of the .npz sample at `path`."""
    f = np.load(path)
    g, d, K, ds = f["gray"], f["depth"], f["K"], float(f["depth_scale"])
    t0 = time.perf_counter()
    for k in range(lo, hi):
        O.gauss_newton(g[k], d[k], g[k + 1], K, ds, levels=levels, max_iterations=iterations, accumulate="fp32")
    return time.perf_counter() - t0, hi - lo


if __name__ == "__main__":      # python cpu_worker.py sample.npz lo hi levels iterations -> prints the seconds of the range
    path, lo, hi, levels, iterations = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    run_pairs(path, lo, min(lo + 1, hi), levels, 1)          # warm-up (imports, page faults) outside the timed range
    print("%.6f" % run_pairs(path, lo, hi, levels, iterations)[0])
