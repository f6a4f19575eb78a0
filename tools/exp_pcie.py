#!/usr/bin/env python3
"""PCIe-inclusive rates: host frames -> device (batched dvo_upload_frames: H2D + pyramid build) and end-to-end."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np
from dvo_amd import synthetic, _cabi
from dvo_amd.batch import BatchAligner, StreamedAligner

B = 512
st = synthetic.make_stream(B + 1, 640, 480, seed=1, device="cuda")
g, d = st["gray"], st["depth"]
gp = _cabi.pinned_array(g.shape, np.uint8); gp[...] = g
dp = _cabi.pinned_array(d.shape, np.uint16); dp[...] = d
al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, max_iterations=10)
for name, (gg, dd) in (("pageable", (g, d)), ("pinned", (gp, dp))):
    al.ingest_host(gg, dd); al.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        al.ingest_host(gg, dd)
        al.ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%s: batched upload+pyramid of %d VGA frames: %.2f ms -> %.0f frames/s, %.1f GB/s over PCIe" % (
        name, B + 1, 1e3 * dt, (B + 1) / dt, (B + 1) * 640 * 480 * 3 / dt / 1e9))
al.close()
for streams in (1, 2, 4):
    sa = StreamedAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, n_streams=streams, max_iterations=10)
    ref = sa.align_host(gp, dp)
    t0 = time.perf_counter()
    for _ in range(5):
        res = sa.align_host(gp, dp)
    dt = (time.perf_counter() - t0) / 5
    print("pinned, %d stream(s): upload + full solve of %d pairs: %.2f ms -> %.0f frame-pairs/s (PCIe-inclusive)" % (streams, B, 1e3 * dt, B / dt))
    sa.close()

# colour input (BGR u8: 3 B/px + depth 2 B/px over the link; conversion + clamp on the device)
bgr = _cabi.pinned_array((B + 1, 480, 640, 3), np.uint8); bgr[...] = g[:, :, :, None]
for streams in (2, 4):
    sa = StreamedAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=B, n_streams=streams, max_iterations=10)
    sa.align_host_bgr(bgr, dp)
    t0 = time.perf_counter()
    for _ in range(5):
        res = sa.align_host_bgr(bgr, dp)
    dt = (time.perf_counter() - t0) / 5
    print("pinned BGR, %d stream(s): upload + colour conversion + full solve of %d pairs: %.2f ms -> %.0f frame-pairs/s (PCIe-inclusive, %.1f GB/s)" % (
        streams, B, 1e3 * dt, B / dt, (B + 1) * 640 * 480 * 5 / dt / 1e9))
    sa.close()
