import sys, time
sys.path.insert(0, "dense-visual-odometry_amd")
import numpy as np
import dvo_amd
from dvo_amd import synthetic, _cabi
st = synthetic.make_stream(12, 640, 480, seed=3, device="cuda")
cam = dvo_amd.RGBDCameraModel(st["K"], st["depth_scale"])
gp = _cabi.pinned_array(st["gray"].shape, np.uint8); gp[...] = st["gray"]
dp = _cabi.pinned_array(st["depth"].shape, np.uint16); dp[...] = st["depth"]
for name, (G, D) in (("pageable", (st["gray"], st["depth"])), ("pinned", (gp, dp))):
    dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=10, solver="device")
    ts = []
    for k in range(36):
        t0 = time.perf_counter()
        dvo.step_gray(G[k % 12], D[k % 12])
        ts.append(time.perf_counter() - t0)
    print(name, "step_gray ms: median %.3f min %.3f" % (1e3 * np.median(ts[1:]), 1e3 * min(ts[1:])))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for k in range(24): dvo.step_gray(G[k % 12], D[k % 12])
    pr.disable()
    if name == "pinned":
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
