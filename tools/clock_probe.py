#!/usr/bin/env python3
"""Clock the chip really holds inside the level-0 reduction kernel (DVFS lowers it under load: MI355X_MICROARCH.md,
'DVFS give-back').  Needs the diagnostic build of the library:

    make -C dense-visual-odometry_amd/csrc OUT=$PWD/ab_libs/libdvo_probe.so EXTRA=-DDVO_CLOCK_PROBE
    DVO_HIP_LIB=$PWD/ab_libs/libdvo_probe.so python tools/clock_probe.py

In that build every workgroup of k_reduce stamps s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its
body and the two reserved slots of the normal-equation record carry their sums.  After >= 2 s of back-to-back launches:
clock = sum(d memtime) / sum(d memrealtime) x 100 MHz."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
import numpy as np  # noqa: E402
import dvo_amd  # noqa: E402
from dvo_amd import synthetic, _cabi  # noqa: E402

B, W, H, L = 256, 640, 480, 4
st = synthetic.make_stream(B + 1, W, H, seed=1000, device="cuda", as_torch=True)
ctx = dvo_amd.HipContext(max_w=W, max_h=H, max_levels=L, n_slots=B + 1, max_pairs=B)
K = np.asarray(st["K"], dtype=np.float32)
ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
slots = np.arange(B + 1, dtype=np.int32)
ctx.upload_frames_device(slots, st["gray"].data_ptr(), st["depth"].data_ptr(), W, H, L)
ctx.synchronize()
T = np.tile(np.eye(4, dtype=np.float32)[:3].reshape(1, 12), (B, 1))
for level in (0, 1):
    t_end = time.time() + 2.5
    n = 0
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    while time.time() < t_end:
        rec = ctx.normal_equations(level, slots[:-1], slots[1:], T)
        n += 1
    launches, ms = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    ticks, real = rec[:, 30].sum(), rec[:, 31].sum()
    if real <= 0:
        raise SystemExit("reserved slots are zero: not the DVO_CLOCK_PROBE build of the library")
    clock_mhz = ticks / real * 100.0
    npts = rec[:, 28].sum()
    us = 1e3 * ms[level] / launches[level]
    print("level %d: %d launches of %d pairs, %.1f us per launch (HIP events); in-kernel clock %.0f MHz "
          "(mean workgroup lifetime %.1f us); %.3g in-bounds points per launch" % (level, n, B, us, clock_mhz, real / rec.shape[0] / (rec[:, 30] > 0).sum() * 0 + real / 100.0 / max(1, int(round(rec[:, 31].size))) / 1.0, npts))
    print("   => %.2f ns per point-wave-instruction slot: at 1024 SIMDs, points/64 x I instructions x c cycles / clock = time" % (us * 1e3 * 1024 * 64 / npts))
ctx.close()
