import sys, os
sys.path.insert(0, "dense-visual-odometry_amd"); sys.path.insert(0, ".")
import numpy as np
import dvo_amd
from dvo_amd import _cabi as cabi
g = np.load("tests/golden/pairs_l4.npz"); f = np.load("tests/golden/frames.npz")
K = f["intrinsics"]
def rel(a, b): return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())
def solve(H, b): return np.linalg.solve(H, b)
with dvo_amd.HipContext(device=0, max_w=640, max_h=480, max_levels=4, n_slots=10, max_pairs=256) as ctx:
    ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], float(f["depth_scale"]))
    for i in range(10): ctx.upload_frame(i, f["gray"][i], f["depth"][i], 4)
    for lvl in range(4):
        idx = np.flatnonzero(g["level"] == lvl); sel = np.resize(idx, 256)
        prev = g["pair"][sel].astype(np.int32)
        rec = ctx.normal_equations(lvl, prev, prev + 1, g["T"][sel])
        es, eh, eb = [], [], []
        for j, i in enumerate(sel[:len(idx)]):
            H, b, swr2, n, sr2 = cabi.unpack_record(rec[j])
            es.append(rel(solve(H, b), g["inc64"][i])); eh.append(rel(H, g["H64"][i])); eb.append(rel(b, g["b64"][i]))
        print("max_groups %s level %d: increment error worst %.2e median %.2e; H worst %.2e; b worst %.2e" % (os.environ.get("DVO_MAX_GROUPS", "default"), lvl, max(es), float(np.median(es)), max(eh), max(eb)))

# which of the two sums carries the increment's error: solve with one of them replaced by the reference run's float64-accumulated sum
with dvo_amd.HipContext(device=0, max_w=640, max_h=480, max_levels=4, n_slots=10, max_pairs=256) as ctx:
    ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], float(f["depth_scale"]))
    for i in range(10): ctx.upload_frame(i, f["gray"][i], f["depth"][i], 4)
    idx = np.flatnonzero(g["level"] == 0); sel = np.resize(idx, 256)
    prev = g["pair"][sel].astype(np.int32)
    rec = ctx.normal_equations(0, prev, prev + 1, g["T"][sel])
    eH, eb = [], []
    for j, i in enumerate(sel[:len(idx)]):
        H, b, swr2, n, sr2 = cabi.unpack_record(rec[j])
        eH.append(rel(solve(H, g["b64"][i]), g["inc64"][i])); eb.append(rel(solve(g["H64"][i], b), g["inc64"][i]))
    print("max_groups %s level 0: increment error with only H from the device worst %.2e, with only b from the device worst %.2e" % (
        os.environ.get("DVO_MAX_GROUPS", "default"), max(eH), max(eb)))
