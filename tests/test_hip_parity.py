"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the golden
vectors of the reference run — on the same inputs.  Run on the GPU box with `pytest -m gpu`.

Tolerances (stated by BASELINE.json north_star): SE(3) increments within 1e-5 relative; everything that is
integer / index work (pyramids, Sobel planes, validity masks, warped coordinates) is bit-exact.
"""
import os

import numpy as np
import pytest
import scipy.linalg

import dvo_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) /
                 max(np.linalg.norm(np.asarray(b, dtype=np.float64)), 1e-300))


def _cabi_unpack(rec):
    from dvo_amd import _cabi
    return _cabi.unpack_record(rec)


def _solve(H, b):
    return scipy.linalg.lstsq(H, b.reshape(6, 1), lapack_driver="gelsy")[0].reshape(6)


@pytest.fixture(scope="module")
def hip(frames):
    """One context holding the ten fixture frames (4 pyramid levels) in slots 0..9."""
    import dvo_amd
    from dvo_amd import _cabi
    ctx = dvo_amd.HipContext(device=0, max_w=640, max_h=480, max_levels=4, n_slots=12, max_pairs=16)
    K = frames["K"]
    ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], frames["depth_scale"])
    for i in range(10):
        ctx.upload_frame(i, frames["gray"][i], frames["depth"][i], 4)
    yield ctx, _cabi
    ctx.close()


@pytest.fixture(scope="module")
def pyramids(frames):
    cache = {}

    def get(k):
        if k not in cache:
            cache[k] = (O.build_pyramid(frames["gray"][k], 4), O.build_pyramid(frames["depth"][k], 4))
        return cache[k]
    return get


def _level(frames, pyramids, k, lvl):
    gp, dp = pyramids(k)
    gc, _ = pyramids(k + 1)
    return O.PairLevel(gp[lvl], dp[lvl], gc[lvl], O.level_intrinsics(frames["K"], lvl), frames["depth_scale"])


# ------------------------------------------------------------------------------------------------
# setup kernels: pyramid (K1) + Sobel pack (K2) — bit exact
# ------------------------------------------------------------------------------------------------
def test_device_pyramid_and_gradients_bit_exact(hip, frames, pyramids):
    ctx, _ = hip
    for slot in (0, 1, 7):
        gp, dp = pyramids(slot)
        for lvl in range(4):
            d = ctx.download_level(slot, lvl)
            assert np.array_equal(d["gray"], gp[lvl])
            assert np.array_equal(d["depth"], dp[lvl])
            gx, gy = O.sobel3(gp[lvl])
            assert np.array_equal(d["gx"].astype(np.float32), gx)
            assert np.array_equal(d["gy"].astype(np.float32), gy)


def test_valid_point_lists_bit_exact(hip, frames, pyramids):
    """The work items of every Gauss-Newton iteration: the depth-valid pixels (camera_model.py:195) of each level in
    raster order with z = (float)((double)depth * depth_scale) (:200-201) and the pixel's intensity — index work, bit exact."""
    ctx, _ = hip
    for k in (0, 7):
        gp, dp = pyramids(k)
        for lvl in range(4):
            got = ctx.download_points(k, lvl)
            ys, xs = np.nonzero(dp[lvl])                                     # row-major == raster order
            assert got["x"].shape[0] == ys.shape[0]
            assert np.array_equal(got["x"], xs) and np.array_equal(got["y"], ys)
            assert np.array_equal(got["intensity"], gp[lvl][ys, xs])
            z = (dp[lvl][ys, xs].astype(np.float64) * frames["depth_scale"]).astype(np.float32)
            assert np.array_equal(got["z"], z)


def test_level_intrinsics_bit_exact(hip, frames):
    ctx, _ = hip
    for lvl in range(5):
        assert np.array_equal(ctx.level_intrinsics(lvl), frames["level_intrinsics"][lvl])


def test_pyramid_odd_and_tiny_sizes():
    import dvo_amd
    rng = np.random.default_rng(3)
    for (h, w, levels) in ((37, 53, 3), (9, 7, 2), (1, 1, 1), (2, 130, 2)):
        gray = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        depth = rng.integers(0, 65536, size=(h, w)).astype(np.uint16)
        with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=levels, n_slots=2, max_pairs=1) as ctx:
            ctx.upload_frame(0, gray, depth, levels)
            gp, dp = O.build_pyramid(gray, levels), O.build_pyramid(depth, levels)
            for lvl in range(levels):
                d = ctx.download_level(0, lvl)
                assert np.array_equal(d["gray"], gp[lvl]) and np.array_equal(d["depth"], dp[lvl])
                gx, gy = O.sobel3(gp[lvl])
                assert np.array_equal(d["gx"], gx.astype(np.int16)) and np.array_equal(d["gy"], gy.astype(np.int16))


def test_bgr_ingest_matches_oracle(frames):
    import dvo_amd
    bgr = frames["color_bgr_1"]
    depth = frames["depth"][0].copy()
    depth[100:120, 200:260] = 40000            # 8 m > max_distance -> zeroed on the device
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=2, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(500, 500, 320, 240, 0, frames["depth_scale"])
        ctx.upload_frame_bgr(0, bgr, depth, 2, 5.0)
        d = ctx.download_level(0, 0, want=("gray", "depth"))
    assert np.array_equal(d["gray"], O.bgr2gray(bgr))
    assert np.array_equal(d["gray"], frames["gray"][0])
    exp = depth.copy()
    exp[(exp * frames["depth_scale"]) > 5.0] = 0
    assert np.array_equal(d["depth"], exp) and (d["depth"][100:120, 200:260] == 0).all()


# ------------------------------------------------------------------------------------------------
# K5 dense writer: validity + warped coordinates bit exact, r / J to float32 rounding
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lvl,tag", [(3, "id"), (3, "xi"), (2, "id"), (2, "xi")])
def test_dense_residuals_jacobian_vs_golden_and_oracle(hip, frames, pyramids, golden_dir, lvl, tag):
    ctx, cabi = hip
    g = np.load(golden_dir / "dense_rj.npz")
    pre = "l%d_%s_" % (lvl, tag)
    T = g[pre + "T"]
    r, J, valid, uv, (h, w) = ctx.residuals_jacobian(lvl, 0, 1, T)
    lv = _level(frames, pyramids, 0, lvl)
    ro, Jo, mask_o, valid_o = O.residuals_and_jacobian(lv, T)
    # integer / index work: bit exact
    assert np.array_equal(valid.reshape(h, w), valid_o)
    assert np.array_equal(~np.isnan(uv[:, 0]).reshape(h, w), g[pre + "mask"])
    x, y, z, m = O.deproject(lv.depth_prev, lv.KL, lv.depth_scale)
    uo, vo = O.project(lv.KL, *O.transform_points(T, x, y, z))
    assert np.array_equal(uv[m.reshape(-1), 0], uo) and np.array_equal(uv[m.reshape(-1), 1], vo)
    # float work
    assert r[valid].shape[0] == g[pre + "r"].shape[0]
    np.testing.assert_allclose(r[valid], g[pre + "r"].reshape(-1), rtol=0, atol=1e-4)
    np.testing.assert_allclose(r[valid], ro.reshape(-1), rtol=0, atol=1e-4)
    scale = np.abs(g[pre + "J"]).max(axis=0)
    np.testing.assert_allclose(J[valid] / scale, g[pre + "J"] / scale, rtol=0, atol=2e-6)
    assert (r[~valid] == 0).all() and (J[~valid] == 0).all()


def test_dense_variant_v2_vs_oracle(hip, frames, pyramids):
    ctx, cabi = hip
    T = O.Pose.from_xi(np.array([0.002, -0.007, -0.011, 0.005, 0.0065, 0.0034])).matrix()
    r, J, valid, uv, (h, w) = ctx.residuals_jacobian(2, 0, 1, T, variant=cabi.VARIANT_V2)
    ro, Jo, _, valid_o = O.residuals_and_jacobian(_level(frames, pyramids, 0, 2), T, "v2")
    assert np.array_equal(valid.reshape(h, w), valid_o)
    np.testing.assert_allclose(r[valid], ro.reshape(-1), rtol=0, atol=1e-4)
    scale = np.abs(Jo).max(axis=0)
    np.testing.assert_allclose(J[valid] / scale, Jo / scale, rtol=0, atol=2e-6)


# ------------------------------------------------------------------------------------------------
# K3 fused reduction: normal equations / increments, teacher-forced with the reference run's poses
# ------------------------------------------------------------------------------------------------
def test_increments_all_pairs_all_iterations_vs_reference_run(hip, frames, golden_dir):
    """All 356 GN iterations of the nine fixture pairs (4 levels x 10 iterations)."""
    ctx, cabi = hip
    g = np.load(golden_dir / "pairs_l4.npz")
    worst = 0.0
    for lvl in range(4):
        idx = np.flatnonzero(g["level"] == lvl)
        for s in range(0, len(idx), 16):
            sel = idx[s:s + 16]
            prev = g["pair"][sel].astype(np.int32)
            rec = ctx.normal_equations(lvl, prev, prev + 1, g["T"][sel])
            for j, i in enumerate(sel):
                H, b, swr2, n, sr2 = cabi.unpack_record(rec[j])
                assert n == int(g["n"][i])                               # validity: bit exact
                inc = _solve(H, b)
                e = _rel(inc, g["inc64"][i])
                worst = max(worst, e)
                assert e < 1e-5, (i, e)
                assert _rel(H, g["H64"][i]) < 1e-6 and _rel(b, g["b64"][i]) < 5e-6
                assert abs(swr2 - g["sum_r2"][i]) <= 1e-6 * g["sum_r2"][i]
    assert worst < 3e-6


@pytest.mark.parametrize("max_groups,bound", [(None, 6e-6), ("16", 3e-6)])
def test_increments_at_the_production_launch_shape_vs_reference_run(frames, golden_dir, monkeypatch, max_groups, bound):
    """The same 356 recorded iterations in launches of 256 pairs - the shape of the throughput path, where a reduction workgroup runs
    through up to 48 record groups (192 records per lane summed in float32 before the float64 stages) instead of the one to four of a
    small batch: validity bit exact, increments within 1e-5 of the reference run's float64-accumulated ones (the longer float32 lane sums
    cost accuracy: worst 4e-6 here against 3e-6 in launches of 16 pairs; DVO_MAX_GROUPS=16 trades 1.2 % of the throughput for the
    accuracy of small launches), and every copy of a
    (pair, pose) in the launch bit-identical to the others."""
    import dvo_amd
    from dvo_amd import _cabi as cabi
    g = np.load(golden_dir / "pairs_l4.npz")
    K = frames["K"]
    worst = 0.0
    if max_groups is None:
        monkeypatch.delenv("DVO_MAX_GROUPS", raising=False)
    else:
        monkeypatch.setenv("DVO_MAX_GROUPS", max_groups)           # (read at dvo_create: include/dvo_hip.h, Environment)
    with dvo_amd.HipContext(device=0, max_w=640, max_h=480, max_levels=4, n_slots=10, max_pairs=256) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], frames["depth_scale"])
        for i in range(10):
            ctx.upload_frame(i, frames["gray"][i], frames["depth"][i], 4)
        for lvl in range(4):
            idx = np.flatnonzero(g["level"] == lvl)
            sel = np.resize(idx, 256)                                    # the level's ~89 records, repeated to fill the launch
            prev = g["pair"][sel].astype(np.int32)
            rec = ctx.normal_equations(lvl, prev, prev + 1, g["T"][sel])
            first = {}
            for j, i in enumerate(sel):
                if i in first:
                    assert np.array_equal(rec[j], rec[first[i]])
                    continue
                first[i] = j
                H, b, swr2, n, sr2 = cabi.unpack_record(rec[j])
                assert n == int(g["n"][i])
                e = _rel(_solve(H, b), g["inc64"][i])
                worst = max(worst, e)
                assert e < 1e-5, (i, e)
                assert _rel(H, g["H64"][i]) < 1e-6 and _rel(b, g["b64"][i]) < 5e-6
    assert worst < bound, worst         # (4.1e-6 at 48 record groups per workgroup, the default; 1.2e-6 at DVO_MAX_GROUPS=16: dvo_abi.hip, Knobs)


def test_increments_single_level_trace(hip, frames, golden_dir):
    ctx, cabi = hip
    g = np.load(golden_dir / "single_l1.npz")
    for i in range(0, len(g["level"]), 3):
        rec = ctx.normal_equations(0, [0], [1], g["T"][i:i + 1])[0]
        H, b, swr2, n, _ = cabi.unpack_record(rec)
        assert n == int(g["n"][i])
        assert _rel(_solve(H, b), g["inc64"][i]) < 1e-5


def test_normal_equations_vs_oracle_random_poses(hip, frames, pyramids):
    ctx, cabi = hip
    rng = np.random.default_rng(11)
    for trial in range(6):
        k = int(rng.integers(0, 9))
        lvl = int(rng.integers(0, 4))
        xi = rng.normal(0, 1, 6) * np.array([0.02, 0.02, 0.02, 0.01, 0.01, 0.01])
        T = O.Pose.from_xi(xi).matrix()
        rec = ctx.normal_equations(lvl, [k], [k + 1], T[None])[0]
        H, b, swr2, n, sr2 = cabi.unpack_record(rec)
        r, J, _, _ = O.residuals_and_jacobian(_level(frames, pyramids, k, lvl), T)
        Ho, bo, so, no = O.normal_equations(r, J)
        assert n == no
        assert _rel(H, Ho) < 1e-6 and _rel(b, bo) < 5e-6 and abs(swr2 - so) < 1e-6 * so
        assert _rel(_solve(H, b), O.solve_increment(Ho, bo)) < 1e-5
        assert abs(swr2 - sr2) <= 1e-8 * sr2


def test_batched_call_equals_single_calls_bitwise(hip, golden_dir):
    ctx, cabi = hip
    g = np.load(golden_dir / "pairs_l4.npz")
    sel = np.flatnonzero(g["level"] == 1)[:12]
    prev = g["pair"][sel].astype(np.int32)
    batch = ctx.normal_equations(1, prev, prev + 1, g["T"][sel])
    for j, i in enumerate(sel):
        one = ctx.normal_equations(1, prev[j:j + 1], prev[j:j + 1] + 1, g["T"][i:i + 1])
        assert np.array_equal(one[0], batch[j])          # fixed-order reduction: deterministic
    again = ctx.normal_equations(1, prev, prev + 1, g["T"][sel])
    assert np.array_equal(again, batch)


def test_student_t_reference_weights(hip, frames, pyramids, golden_dir):
    ctx, cabi = hip
    g = np.load(golden_dir / "weighted_l4.npz")
    for i in range(0, len(g["level"]), 3):
        lvl = int(g["level"][i])
        T = g["T"][i:i + 1]
        r, J, _, _ = O.residuals_and_jacobian(_level(frames, pyramids, 0, lvl), T[0])
        r2 = (r * r).astype(np.float64).reshape(-1)
        # scale pass (weighter/t_weighter.py:36-47)
        for lam in (1.0 / 25.0, 1.0 / O.student_t_scale_sum(r2, 1.0 / 25.0)):
            s = ctx.weighted_r2_sum(lvl, [0], [1], T, [lam])[0]
            assert abs(s - O.student_t_scale_sum(r2, lam)) <= 2e-6 * s
        w, lam = O.student_t_weights(r)
        rec = ctx.normal_equations(lvl, [0], [1], T, cabi.VARIANT_V1, cabi.WEIGHT_STUDENT_T, [lam])[0]
        H, b, swr2, n, sr2 = cabi.unpack_record(rec)
        assert n == int(g["n"][i])
        assert _rel(_solve(H, b), g["inc64"][i]) < 1e-5
        assert abs(swr2 - g["sum_wr2"][i]) <= 2e-6 * g["sum_wr2"][i]


def test_variant_v2_increments(hip, golden_dir):
    ctx, cabi = hip
    g = np.load(golden_dir / "approx_l4.npz")
    for i in range(0, len(g["level"]), 3):
        rec = ctx.normal_equations(int(g["level"][i]), [0], [1], g["T"][i:i + 1], cabi.VARIANT_V2)[0]
        H, b, _, n, _ = cabi.unpack_record(rec)
        assert n == int(g["n"][i])
        assert _rel(_solve(H, b), g["inc64"][i]) < 1e-5


def test_huber_weights_vs_oracle(hip, frames, pyramids):
    """Build-defined weight (the reference has no Huber): parity against the oracle only."""
    ctx, cabi = hip
    T = np.eye(4, dtype=np.float32)
    r, J, _, _ = O.residuals_and_jacobian(_level(frames, pyramids, 2, 1), T)
    k = 1.345 * float(np.sqrt(np.mean(r.astype(np.float64) ** 2)))
    Ho, bo, so, no = O.normal_equations(r, J, O.huber_weights(r, k))
    rec = ctx.normal_equations(1, [2], [3], T[None], cabi.VARIANT_V1, cabi.WEIGHT_HUBER, [k])[0]
    H, b, swr2, n, sr2 = cabi.unpack_record(rec)
    assert n == no and _rel(H, Ho) < 1e-6 and _rel(_solve(H, b), O.solve_increment(Ho, bo)) < 1e-5
    assert abs(swr2 - so) < 1e-6 * so


# ------------------------------------------------------------------------------------------------
# full Gauss-Newton: host-solve mode (north-star literal) and device-solve mode
# ------------------------------------------------------------------------------------------------
def _make_dvo(frames, **kw):
    import dvo_amd
    cam = dvo_amd.RGBDCameraModel(frames["K"], frames["depth_scale"])
    return dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), **kw), dvo_amd


@pytest.mark.parametrize("solver", ["host", "device"])
def test_step_api_four_levels_vs_reference_run(frames, golden_dir, solver):
    g = np.load(golden_dir / "pairs_l4.npz")
    dvo, _ = _make_dvo(frames, levels=4, max_iterations=10, solver=solver)
    for k in (0, 1):
        c = np.repeat(frames["gray"][k][..., None], 3, axis=2)
        tf = dvo.step(c, frames["depth"][k].copy())
        if k == 0:
            np.testing.assert_array_equal(tf.exp(), np.eye(4, dtype=np.float32))
    assert dvo.last_iterations == 40
    np.testing.assert_allclose(tf.log().reshape(6), g["final_xi"][0], rtol=0, atol=5e-7)
    np.testing.assert_allclose(tf.exp(), g["final_T"][0], rtol=0, atol=5e-7)


@pytest.mark.parametrize("solver", ["host", "device"])
def test_step_with_init_guess_vs_oracle(frames, solver):
    """step(..., init_guess=T) (core/base_dense_visual_odometry.py:54-69): the estimate of pair (1, 2) started from the
    transform found for pair (0, 1), against the oracle started from the same pose."""
    dvo, _ = _make_dvo(frames, levels=3, max_iterations=6, solver=solver)
    dvo.step_gray(frames["gray"][0], frames["depth"][0])
    tf1 = dvo.step_gray(frames["gray"][1], frames["depth"][1])
    tf2 = dvo.step_gray(frames["gray"][2], frames["depth"][2], init_guess=tf1)
    est = O.gauss_newton(frames["gray"][1], frames["depth"][1], frames["gray"][2], frames["K"], frames["depth_scale"],
                         levels=3, max_iterations=6, init=O.Pose.from_xi(tf1.log().reshape(6)))
    assert dvo.last_iterations == 18
    np.testing.assert_allclose(tf2.log().reshape(6), est.log(), rtol=0, atol=3e-6)
    assert np.abs(tf2.log().reshape(6) - tf1.log().reshape(6)).max() > 1e-4          # it did move away from the guess


@pytest.mark.parametrize("solver", ["host", "device"])
def test_single_level_to_convergence(frames, golden_dir, solver):
    """Free-running stop rules (|d err| < tolerance, error increase) of base_robust_dvo.py:201-232: the reference run
    stops after 54 iterations at a single level; so do the host loop and the device-side control kernel."""
    g = np.load(golden_dir / "single_l1.npz")
    dvo, _ = _make_dvo(frames, levels=1, solver=solver)
    dvo.step_gray(frames["gray"][0], frames["depth"][0])
    tf = dvo.step_gray(frames["gray"][1], frames["depth"][1])
    assert dvo.last_iterations == len(g["level"])
    np.testing.assert_allclose(tf.log().reshape(6), g["final_xi"], rtol=0, atol=5e-7)


def test_device_solve_batch_nine_pairs(hip, frames, pyramids, golden_dir):
    """dvo_solve_batch: nine pairs in one batch, stop rules per pair on the device (pair 7 stops early)."""
    ctx, cabi = hip
    g = np.load(golden_dir / "pairs_l4.npz")
    prev = np.arange(9, dtype=np.int32)
    ctx.trace_enable(40)
    res = ctx.solve_batch(prev, prev + 1, levels=4, max_iterations=10)
    for k in range(9):
        assert res["iters"][k] == int((g["pair"] == k).sum())
        np.testing.assert_allclose(res["xi"][k], g["final_xi"][k], rtol=0, atol=5e-7)
        np.testing.assert_allclose(res["T"][k], g["final_T"][k][:3], rtol=0, atol=2e-6)   # golden R carries the float32 rounding of So3.exp
    # per-iteration trace: increments of the free-running device loop vs the reference run's
    tr = ctx.trace_read(9)
    for k in range(9):
        sel = np.flatnonzero(g["pair"] == k)
        for j, i in enumerate(sel):
            assert int(round(tr[k, j, 28])) == int(g["n"][i])
            # free-running: the two runs' poses drift apart by ~1e-8, which is ~1e-4 of a late (small) increment
            assert np.abs(tr[k, j, 32:38] - g["inc64"][i]).max() < 5e-8
    # teacher-forced: the oracle evaluated at exactly the pose the device used for that iteration
    for k in (0, 7):
        sel = np.flatnonzero(g["pair"] == k)
        for j, i in enumerate(sel):
            T = np.eye(4, dtype=np.float32)
            T[:3, :] = tr[k, j, 38:50].reshape(3, 4).astype(np.float32)
            r, J, _, _ = O.residuals_and_jacobian(_level(frames, pyramids, k, int(g["level"][i])), T)
            Ho, bo, so, no = O.normal_equations(r, J)
            assert no == int(round(tr[k, j, 28]))
            assert _rel(tr[k, j, 32:38], O.solve_increment(Ho, bo)) < 1e-5
    ctx.trace_enable(0)


def test_device_solve_student_t_and_v2(hip, golden_dir):
    ctx, cabi = hip
    gw = np.load(golden_dir / "weighted_l4.npz")
    res = ctx.solve_batch([0], [1], levels=4, max_iterations=10, weight_mode=cabi.WEIGHT_STUDENT_T)
    assert res["iters"][0] == len(gw["level"])
    np.testing.assert_allclose(res["xi"][0], gw["final_xi"], rtol=0, atol=1e-6)
    ga = np.load(golden_dir / "approx_l4.npz")
    res = ctx.solve_batch([0], [1], levels=4, max_iterations=10, variant=cabi.VARIANT_V2)
    assert res["iters"][0] == len(ga["level"])
    np.testing.assert_allclose(res["xi"][0], ga["final_xi"], rtol=0, atol=1e-6)


def test_device_solve_student_t_one_pass_iterations_vs_reference_run(hip, frames, pyramids, golden_dir):
    """The DEFAULT Student-t solve (one pass per iteration at 640x480 and 320x240: corrected / predicted-lambda forms; the reference's
    loop as written inside the persistent kernel at 160x120 and 80x60) ITERATION BY ITERATION against the reference run of
    use_weighter=True (weighted_l4.npz: base_robust_dvo.py:171-188 with weighter/t_weighter.py:21-47): every traced increment
    against the run's float64 increment (free-running, absolute) and, teacher-forced at the pose the device evaluated, against
    the oracle's weighter loop at the north-star tolerance of 1e-5 relative.  No pair is solved again, and every one-pass
    iteration's own bound (trace slot 30) is inside the 5e-6 it is kept under."""
    ctx, cabi = hip
    gw = np.load(golden_dir / "weighted_l4.npz")
    n_it = len(gw["level"])
    fb0 = ctx.student_t_fallback_pairs()
    ctx.trace_enable(n_it)
    res = ctx.solve_batch([0], [1], levels=4, max_iterations=10, weight_mode=cabi.WEIGHT_STUDENT_T)
    tr = ctx.trace_read(1)
    ctx.trace_enable(0)
    assert res["iters"][0] == n_it
    assert ctx.student_t_fallback_pairs() == fb0
    worst_rel, worst_bound = 0.0, 0.0
    for j in range(n_it):
        lvl = int(gw["level"][j])
        assert int(round(tr[0, j, 28])) == int(gw["n"][j])
        assert np.abs(tr[0, j, 32:38] - gw["inc64"][j]).max() < 5e-8
        np.testing.assert_allclose(tr[0, j, 27], gw["sum_wr2"][j], rtol=1e-5)         # (free-running; feeds a float32 mean)
        T = np.eye(4, dtype=np.float32)
        T[:3, :] = tr[0, j, 38:50].reshape(3, 4).astype(np.float32)
        r, J, _, _ = O.residuals_and_jacobian(_level(frames, pyramids, 0, lvl), T)
        w, _ = O.student_t_weights(r)
        Ho, bo, so, no = O.normal_equations(r, J, w)
        rel = _rel(tr[0, j, 32:38], O.solve_increment(Ho, bo))
        worst_rel = max(worst_rel, rel)
        assert rel < 1e-5, (j, lvl, rel)
        if lvl <= 1:                                              # one-pass forms: the iteration's own bound, relative to its increment
            assert 0.0 < tr[0, j, 30] <= 5e-6, (j, lvl, tr[0, j, 30])
            worst_bound = max(worst_bound, tr[0, j, 30])
    print("one-pass Student-t vs oracle, teacher-forced: worst relative increment error %.2e; worst kept bound %.2e" % (worst_rel, worst_bound))


@pytest.mark.parametrize("solver", ["host", "device"])
def test_rank_rule_of_the_solve_vs_reference_runs(golden_dir, solver):
    """a15 on both solve paths: the three reference runs of rank_cases.npz (cond(H) = inf / 1.5e8 / 1.6e3; see the CPU test of the same
    name).  Every iteration's increment against the reference's rank rule (float32 eps) applied to the oracle's float64 normal
    equations at the pose the solve used, the first one also against the reference run's own numbers."""
    import dvo_amd
    g = np.load(golden_dir / "rank_cases.npz")
    K, scale, depth = g["intrinsics"], float(g["depth_scale"]), g["depth"]
    with dvo_amd.HipContext(device=0, max_w=160, max_h=120, max_levels=1, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], scale)
        for name in ("stripes", "window", "textured"):
            g1, g2 = g[name + "_gray1"], g[name + "_gray2"]
            ctx.upload_frame(0, g1, depth, 1)
            ctx.upload_frame(1, g2, depth, 1)
            lv = O.PairLevel(g1, depth, g2, O.level_intrinsics(K, 0), scale)
            if solver == "device":
                ctx.trace_enable(3)
                res = ctx.solve_batch([0], [1], levels=1, max_iterations=3)
                tr = ctx.trace_read(1)
                ctx.trace_enable(0)
                incs = [tr[0, j, 32:38] for j in range(int(res["iters"][0]))]
                poses = [tr[0, j, 38:50].reshape(3, 4) for j in range(int(res["iters"][0]))]
                xi = res["xi"][0]
            else:
                incs, poses, T = [], [], np.eye(4, dtype=np.float32)
                est = O.Pose.identity()
                for j in range(3):
                    rec = ctx.normal_equations(0, [0], [1], est.matrix()[None])[0]
                    H, b, _, _, _ = _cabi_unpack(rec)
                    inc = scipy.linalg.lstsq(H, b.reshape(6, 1), cond=float(np.finfo(np.float32).eps), lapack_driver="gelsy")[0].reshape(6)
                    incs.append(inc)
                    poses.append(est.matrix()[:3])
                    est = O.Pose.from_xi(inc).mul(est)
                xi = est.log()
            assert len(incs) == 3
            for j, (inc, P) in enumerate(zip(incs, poses)):
                T = np.eye(4, dtype=np.float32)
                T[:3] = np.asarray(P, dtype=np.float32)
                r, J, _, _ = O.residuals_and_jacobian(lv, T)
                Ho, bo, _, _ = O.normal_equations(r, J)
                assert _rel(inc, O.solve_increment(Ho, bo)) < 1e-5, (name, solver, j)
                if name != "textured":
                    assert abs(inc[1]) < 1e-8, (name, solver, j, inc)              # the unconstrained direction carries no step
            assert _rel(incs[0], g[name + "_inc32"][0]) < 5e-4                     # the reference run's float32 solve, to its rounding
            np.testing.assert_allclose(xi, g[name + "_final_xi"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("solver", ["host", "device"])
def test_motion_prior_sigma_vs_reference_run(frames, golden_dir, solver):
    """sigma prior (base_robust_dvo.py:190-194,:217-218) through step(): iteration counts and poses of the
    reference run (which itself carries ~1e-6 of float32 rounding from adding 1e13 to a float32 H)."""
    g = np.load(golden_dir / "sigma_l4.npz")
    dvo, _ = _make_dvo(frames, levels=4, max_iterations=10, sigma=float(g["sigma"]), solver=solver)
    for k in range(4):
        c = np.repeat(frames["gray"][k][..., None], 3, axis=2)
        tf = dvo.step(c, frames["depth"][k].copy())
        if k:
            assert dvo.last_iterations == int((g["pair"] == k - 1).sum())
            np.testing.assert_allclose(tf.log().reshape(6), g["final_xi"][k - 1], rtol=0, atol=5e-6)


def test_sequence_pose_chain_vs_reference_run(frames, golden_dir):
    """step() over the ten fixture frames: current_pose chain (base_dense_visual_odometry.py:79)."""
    import dvo_amd
    g = np.load(golden_dir / "sequence.npz")
    cam = dvo_amd.RGBDCameraModel(frames["K"], frames["depth_scale"])
    T0 = frames["gt_T"][0]
    init = dvo_amd.Se3(dvo_amd.So3(T0[:3, :3]), T0[:3, 3].reshape(3, 1))
    dvo = dvo_amd.get_dvo("robust-dvo", cam, init, levels=4, max_iterations=10)
    for k in range(10):
        c = np.repeat(frames["gray"][k][..., None], 3, axis=2)
        tf = dvo.step(c, frames["depth"][k].copy())
        np.testing.assert_allclose(tf.exp(), g["tf_T"][k], rtol=0, atol=1e-6)
        np.testing.assert_allclose(dvo.current_pose.exp(), g["pose_T"][k], rtol=0, atol=5e-6)


# ------------------------------------------------------------------------------------------------
# the reference's own tests for this path, through the drop-in API
# ------------------------------------------------------------------------------------------------
def test_reference_known_answer_10x10(golden_dir):
    # reference test_cpu_robust_dense_visual_odometry.py:20-44
    import dvo_amd
    height, width = 10, 10
    gray = np.full((height, width), 150, dtype=np.uint8)
    gray[:5, :5] = 150 / 3.0
    color = np.full((height, width, 3), 150, dtype=np.uint8)
    color[:5, :5, :] = 150 / 3.0
    depth = np.ones_like(gray).astype(np.uint16)
    depth[:5, :5] = 3
    cam = dvo_amd.RGBDCameraModel(np.eye(3, dtype=np.float32), 1.0)
    dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=1, height=10, width=10)
    dvo.step(color_image=color, depth_image=depth)
    dvo._build_pyramids(gray_image=gray, depth_image=depth)
    dvo._setup(level=0)
    residuals, J, mask = dvo.compute_residuals_and_jacobian(estimate=dvo_amd.Se3.identity(), level=0)
    np.testing.assert_almost_equal(residuals, np.zeros((100, 1), dtype=np.float32))
    t = np.load(golden_dir / "tiny10.npz")
    assert J.shape == (100, 6) and mask.all()
    np.testing.assert_allclose(J, t["J"], rtol=1e-6, atol=1e-6)
    # a constant image gives H = 0: gelsy's minimum-norm answer is a zero increment, on host and device
    dvo2 = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=1, height=10, width=10, solver="device")
    flat = np.full((10, 10, 3), 90, dtype=np.uint8)
    dvo2.step(flat, depth.copy())
    tf = dvo2.step(flat, depth.copy())
    np.testing.assert_array_equal(tf.log(), np.zeros((6, 1), dtype=np.float32))


@pytest.mark.parametrize("idx", [4, 5])
def test_reference_ground_truth_pose_residuals(frames, idx):
    # reference test_cpu_robust_dense_visual_odometry.py:46-73
    import dvo_amd
    cam = dvo_amd.RGBDCameraModel(frames["K"], frames["depth_scale"])
    T0, T1 = frames["gt_T"][idx - 1], frames["gt_T"][idx]
    s0 = dvo_amd.Se3(dvo_amd.So3(T0[:3, :3]), T0[:3, 3].reshape(3, 1))
    s1 = dvo_amd.Se3(dvo_amd.So3(T1[:3, :3]), T1[:3, 3].reshape(3, 1))
    gt = s1.inverse() * s0
    dvo = dvo_amd.get_dvo("robust-dvo", cam, s0, levels=1)
    dvo.step_gray(frames["gray"][idx - 1], frames["depth"][idx - 1])
    dvo._build_pyramids(gray_image=frames["gray"][idx], depth_image=frames["depth"][idx])
    dvo._setup(level=0)
    residuals, _, _ = dvo.compute_residuals_and_jacobian(estimate=gt, level=0)
    assert np.isclose(residuals.mean(), 0.0, atol=5.0)
    assert residuals.std() < 20


# ------------------------------------------------------------------------------------------------
# error behaviour of the boundary
# ------------------------------------------------------------------------------------------------
def test_error_codes(hip, frames):
    ctx, cabi = hip
    import dvo_amd
    T = np.eye(4, dtype=np.float32)[None]
    with pytest.raises(dvo_amd.DVOHipError) as e:
        ctx.normal_equations(0, [0], [11], T)              # slot 11 never uploaded
    assert e.value.code == -5
    with pytest.raises(dvo_amd.DVOHipError) as e:
        ctx.normal_equations(4, [0], [1], T)               # level not built
    assert e.value.code == -1
    with pytest.raises(dvo_amd.DVOHipError) as e:
        ctx.normal_equations(0, [0], [1], T, variant=7)
    assert e.value.code == -1
    with pytest.raises(dvo_amd.DVOHipError):
        ctx.normal_equations(0, [0] * 17, [1] * 17, np.repeat(T, 17, axis=0))      # > max_pairs
    with pytest.raises(dvo_amd.DVOHipError):
        ctx.upload_frame(0, np.zeros((481, 640), np.uint8), np.zeros((481, 640), np.uint16), 4)
    ctx.upload_frame(0, frames["gray"][0], frames["depth"][0], 4)                     # restore slot 0
    with dvo_amd.HipContext(max_w=64, max_h=48, max_levels=2, n_slots=2, max_pairs=1) as c2:
        c2.upload_frame(0, np.zeros((48, 64), np.uint8), np.ones((48, 64), np.uint16), 2)
        c2.upload_frame(1, np.zeros((48, 64), np.uint8), np.ones((48, 64), np.uint16), 2)
        with pytest.raises(dvo_amd.DVOHipError) as e:
            c2.normal_equations(0, [0], [1], T)            # camera not set
        assert e.value.code == -5


def test_all_depth_invalid_gives_empty_system(frames):
    import dvo_amd
    from dvo_amd import _cabi
    with dvo_amd.HipContext(max_w=64, max_h=48, max_levels=1, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(60, 60, 32, 24, 0, 0.001)
        g = np.random.default_rng(0).integers(0, 255, size=(48, 64), dtype=np.uint8)
        ctx.upload_frame(0, g, np.zeros((48, 64), np.uint16), 1)
        ctx.upload_frame(1, g, np.zeros((48, 64), np.uint16), 1)
        rec = ctx.normal_equations(0, [0], [1], np.eye(4, dtype=np.float32)[None])[0]
        assert (rec == 0).all()
        res = ctx.solve_batch([0], [1], levels=1, max_iterations=5)
        assert res["iters"][0] == 1 and np.all(res["xi"][0] == 0)


# ------------------------------------------------------------------------------------------------
# BASELINE full sizes: size-independent properties (the oracle is too slow to sweep these)
# ------------------------------------------------------------------------------------------------
def _synthetic_frame(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w), np.float32)
    for o in range(5):
        f = 2.0 ** o
        img += rng.uniform(0.3, 1.0) * np.sin(xx * f * 0.011 + rng.uniform(0, 6)) * np.cos(yy * f * 0.013 + rng.uniform(0, 6)) / f ** 0.5
    img = (255 * (img - img.min()) / (img.max() - img.min())).astype(np.uint8)
    depth = (5000 + 3000 * np.sin(xx * 0.004) * np.cos(yy * 0.005) + 4000).astype(np.uint16)
    depth[rng.random((h, w)) < 0.2] = 0
    depth[:16], depth[-16:], depth[:, :16], depth[:, -16:] = 0, 0, 0, 0
    return img, depth


def test_full_size_1280x960_properties():
    import dvo_amd
    from dvo_amd import _cabi
    rng = np.random.default_rng(1234)
    h, w, levels = 960, 1280, 5
    img, depth = _synthetic_frame(rng, h, w)
    with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=levels, n_slots=4, max_pairs=4) as ctx:
        ctx.set_camera(1034.6, 1033.0, 637.2, 479.0, 0.0, 0.0002)
        ctx.upload_frame(0, img, depth, levels)
        ctx.upload_frame(1, img, depth, levels)
        I = np.eye(4, dtype=np.float32)[None]
        for lvl in range(levels):
            # (a) identical frames + identity pose: zero residual, zero gradient of the cost, N = valid count
            rec = ctx.normal_equations(lvl, [0], [1], I)[0]
            H, b, swr2, n, sr2 = _cabi.unpack_record(rec)
            d = ctx.download_level(0, lvl, want=("depth",))["depth"]
            r, J, valid, uv, _ = ctx.residuals_jacobian(lvl, 0, 1, I[0])
            assert n == int(valid.sum()) <= int((d != 0).sum())
            # (coordinates round to ~1e-4 px at the identity, so residuals are tiny, not exactly zero)
            assert swr2 < 1e-6 * max(n, 1)
            assert np.all(np.abs(b) <= np.sqrt(np.diag(H) * swr2) * (1 + 1e-9))       # Cauchy-Schwarz
            # (b) the fused reduction equals the float64 sum of the dense kernel's rows
            J64 = J[valid].astype(np.float64)
            assert _rel(H, J64.T @ J64) < 1e-6
            # (c) H symmetric positive semi-definite
            assert np.linalg.eigvalsh(H).min() > -1e-9 * np.abs(H).max()
        # (d) weights: student-t with lambda -> 0 is 1.2 x the unweighted system; huber with huge k is unweighted
        T = O.Pose.from_xi(np.array([0.004, -0.003, 0.002, 0.002, -0.001, 0.003])).matrix()[None]
        r0 = ctx.normal_equations(0, [0], [1], T)[0]
        rt = ctx.normal_equations(0, [0], [1], T, _cabi.VARIANT_V1, _cabi.WEIGHT_STUDENT_T, [0.0])[0]
        rh = ctx.normal_equations(0, [0], [1], T, _cabi.VARIANT_V1, _cabi.WEIGHT_HUBER, [1e30])[0]
        assert _rel(rt[:21], 1.2 * r0[:21]) < 1e-6 and _rel(rt[21:27], 1.2 * r0[21:27]) < 1e-5
        assert abs(rt[27] - 1.2 * r0[27]) < 1e-6 * rt[27] and rt[28] == r0[28]
        np.testing.assert_allclose(rh[:28], r0[:28], rtol=1e-12)
        # (e) a solve from a perturbed start on identical frames walks back towards the identity
        res = ctx.solve_batch([0], [1], levels=levels, max_iterations=10, T_init=T)
        assert np.linalg.norm(res["xi"][0]) < 0.2 * np.linalg.norm([0.004, -0.003, 0.002, 0.002, -0.001, 0.003])


@pytest.mark.parametrize("hw", [(1, 1), (2, 3), (5, 4), (2, 16), (17, 33), (3, 130)])
def test_tiny_and_ragged_images_lists_and_normal_equations(hw):
    """Degenerate geometries (single pixel, fewer pixels than a wave, one partly filled list segment, rows shorter than
    a tile): valid-point lists bit exact, validity count and normal equations against the oracle."""
    import dvo_amd
    from dvo_amd import _cabi
    h, w = hw
    rng = np.random.default_rng(h * 1000 + w)
    g1 = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    g2 = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    depth = rng.integers(3000, 9000, size=(h, w)).astype(np.uint16)
    depth[rng.random((h, w)) < 0.3] = 0
    K = np.array([[1.1 * max(w, 2), 0, (w - 1) / 2], [0, 1.1 * max(w, 2), (h - 1) / 2], [0, 0, 1]], dtype=np.float32)
    T = O.Pose.from_xi(np.array([0.001, -0.002, 0.001, 0.002, 0.001, -0.003])).matrix()
    with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=1, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.0, 0.0002)
        ctx.upload_frame(0, g1, depth, 1)
        ctx.upload_frame(1, g2, depth, 1)
        pts = ctx.download_points(0, 0)
        ys, xs = np.nonzero(depth)
        assert np.array_equal(pts["x"], xs) and np.array_equal(pts["y"], ys) and np.array_equal(pts["intensity"], g1[ys, xs])
        lv = O.PairLevel(g1, depth, g2, O.level_intrinsics(K, 0), 0.0002)
        ro, Jo, _, valid_o = O.residuals_and_jacobian(lv, T, "v1")
        Ho, bo, so, no = O.normal_equations(ro, Jo)
        H, b, swr2, n, sr2 = _cabi.unpack_record(ctx.normal_equations(0, [0], [1], T[None])[0])
        assert n == no == int(valid_o.sum())
        if no:
            assert _rel(H, Ho) < 1e-5 and abs(swr2 - so) <= 1e-5 * max(so, 1e-9)
        else:
            assert not H.any() and swr2 == 0.0


def test_maximum_image_size_properties():
    """The largest geometry the boundary accepts (4096 wide, (h + 1) * pitch just below 2^24: 12-bit coordinates in the
    valid-point records, float32 tap offsets): lists against NumPy, fused reduction against the dense kernel's rows, and
    one size further is refused."""
    import dvo_amd
    from dvo_amd import _cabi
    rng = np.random.default_rng(99)
    w, h, levels = 4096, 4078, 2                               # pitch 4112: 4079 * 4112 = 16 772 848 < 2^24
    img, depth = _synthetic_frame(rng, h, w)
    with pytest.raises(_cabi.DVOHipError):
        dvo_amd.HipContext(max_w=4096, max_h=4080, max_levels=1, n_slots=2, max_pairs=1)
    with pytest.raises(_cabi.DVOHipError):
        dvo_amd.HipContext(max_w=4112, max_h=1000, max_levels=1, n_slots=2, max_pairs=1)
    with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=levels, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(3300.0, 3300.0, 2040.3, 2031.7, 0.0, 0.0002)
        ctx.upload_frame(0, img, depth, levels)
        ctx.upload_frame(1, img, depth, levels)
        pts = ctx.download_points(0, 0)
        ys, xs = np.nonzero(depth)
        assert np.array_equal(pts["x"], xs) and np.array_equal(pts["y"], ys)      # x up to 4095, y up to 4077
        assert np.array_equal(pts["intensity"], img[ys, xs])
        T = O.Pose.from_xi(np.array([0.003, -0.002, 0.001, 0.001, -0.0005, 0.002])).matrix()
        for lvl in range(levels):
            rec = ctx.normal_equations(lvl, [0], [1], T[None])[0]
            H, b, swr2, n, sr2 = _cabi.unpack_record(rec)
            r, J, valid, uv, _ = ctx.residuals_jacobian(lvl, 0, 1, T)
            assert n == int(valid.sum())
            J64, r64 = J[valid].astype(np.float64), r[valid].astype(np.float64)
            assert _rel(H, J64.T @ J64) < 1e-6 and _rel(b, -(J64.T @ r64)) < 1e-5
            assert abs(swr2 - float(r64 @ r64)) < 1e-6 * swr2


def test_levels_with_and_without_float_pattern_tap_offsets_vs_oracle():
    """2048 x 1100: level 0 is past the size up to which tap byte offsets are formed as float32 bit patterns (4 (h + 1) pitch >= 2^23:
    offsets by conversion and shift, the general instantiations), levels 1 and 2 are below it (the fast ones) - one context, one solve,
    both kinds of launch.  Each level's validity count and normal equations against the oracle, and the device solve (3 iterations per
    level) against the oracle's."""
    import dvo_amd
    from dvo_amd import _cabi
    rng = np.random.default_rng(4242)
    w, h, levels = 2048, 1100, 3
    img, depth = _synthetic_frame(rng, h, w)
    img2 = np.roll(img, (1, 2), axis=(0, 1))
    K = np.array([[1650.0, 0, 1023.5], [0, 1650.0, 549.5], [0, 0, 1]], dtype=np.float32)
    T = O.Pose.from_xi(np.array([0.002, -0.001, 0.001, 0.001, -0.0005, 0.001])).matrix()
    with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=levels, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.0, 0.0002)
        ctx.upload_frame(0, img, depth, levels)
        ctx.upload_frame(1, img2, depth, levels)
        pg1, pd1 = O.build_pyramid(img, levels), O.build_pyramid(depth, levels)
        pg2 = O.build_pyramid(img2, levels)
        for lvl in range(levels):
            lv = O.PairLevel(pg1[lvl], pd1[lvl], pg2[lvl], O.level_intrinsics(K, lvl), 0.0002)
            ro, Jo, _, valid_o = O.residuals_and_jacobian(lv, T, "v1")
            Ho, bo, so, no = O.normal_equations(ro, Jo)
            H, b, swr2, n, sr2 = _cabi.unpack_record(ctx.normal_equations(lvl, [0], [1], T[None])[0])
            assert n == no == int(valid_o.sum()), lvl
            assert _rel(H, Ho) < 1e-5 and _rel(b, bo) < 1e-4 and abs(swr2 - so) <= 1e-5 * so, lvl
        dev = ctx.solve_batch([0], [1], levels=levels, max_iterations=3)
    est = O.gauss_newton(img, depth, img2, K, 0.0002, levels, max_iterations=3)
    assert int(dev["iters"][0]) == 3 * levels
    np.testing.assert_allclose(dev["xi"][0], est.log(), rtol=0, atol=1e-4 * np.abs(est.log()).max())       # (3e-6 on the fixture pairs is 3e-4 of their xi)


def test_persistent_kernel_over_levels_with_and_without_float_pattern_tap_offsets(monkeypatch):
    """The persistent per-pair kernel serves all its levels with ONE instantiation: when one of them is too large for float-pattern tap
    offsets (2048 x 1100 at level 0) the general one, also for the smaller levels, whose geometry carries the bias.  Every level forced
    into that kernel (DVO_SMALL_MAX_PIX) against the per-iteration launches of the default configuration."""
    import dvo_amd
    rng = np.random.default_rng(4243)
    w, h, levels = 2048, 1100, 3
    img, depth = _synthetic_frame(rng, h, w)
    img2 = np.roll(img, (1, 2), axis=(0, 1))
    res = []
    for small in (None, "100000000"):
        if small is None:
            monkeypatch.delenv("DVO_SMALL_MAX_PIX", raising=False)
        else:
            monkeypatch.setenv("DVO_SMALL_MAX_PIX", small)
        with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=levels, n_slots=2, max_pairs=1) as ctx:
            ctx.set_camera(1650.0, 1650.0, 1023.5, 549.5, 0.0, 0.0002)
            ctx.upload_frame(0, img, depth, levels)
            ctx.upload_frame(1, img2, depth, levels)
            res.append(ctx.solve_batch([0], [1], levels=levels, max_iterations=2))
    assert np.array_equal(res[0]["iters"], res[1]["iters"]) and int(res[0]["iters"][0]) == 2 * levels
    np.testing.assert_allclose(res[1]["xi"][0], res[0]["xi"][0], rtol=0, atol=1e-5 * np.abs(res[0]["xi"]).max())


def test_projection_division_is_bit_identical_to_ieee(hip):
    """The projection's quotient (shared refined reciprocal + ONE residual correction, no v_div_scale / v_div_fixup) against the
    compiler's IEEE division on the device and against NumPy, over the operand ranges the projection sees."""
    ctx, _ = hip
    rng = np.random.default_rng(2024)
    n = 1 << 22
    for lo, hi, scale in ((0.05, 20.0, 3000.0), (0.3, 6.0, 800.0), (1e-3, 1e3, 1e6)):
        b = rng.uniform(lo, hi, n).astype(np.float32)
        a = (rng.uniform(-1.0, 1.0, n) * scale * b).astype(np.float32)
        out, ref = ctx.debug_divide(a, b)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(out, a / b)
    # divisors whose mantissa is all ones, or all ones but the last bits (the case Markstein's theorem on the one-correction quotient
    # excepts), against numerators of every kind; tools/ubench/quotient3.hip sweeps every mantissa on the device
    m = np.concatenate([np.full(1 << 18, 0x7FFFFF, dtype=np.uint32), rng.integers(0x7FFFF0, 0x800000, 1 << 18, dtype=np.uint32)])
    e = rng.integers(120, 132, m.size, dtype=np.uint32)
    b = ((e << 23) | m).view(np.float32)
    a = (rng.uniform(-1.0, 1.0, m.size) * 3000.0 * b).astype(np.float32)
    a[::7] = (b[::7] * np.float32(1.9999999)).astype(np.float32)
    out, ref = ctx.debug_divide(a, b)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)) and np.array_equal(out, a / b)
    # negative / tiny denominators keep the same answer too; zero gives a non-finite value (pixel invalid either way)
    b = np.array([-2.5, -0.01, 1e-20, 0.0], dtype=np.float32)
    a = np.array([100.0, 3.0, 1e-18, 5.0], dtype=np.float32)
    out, ref = ctx.debug_divide(a, b)
    assert np.array_equal(out[:3], ref[:3]) and not np.isfinite(out[3]) and not np.isfinite(ref[3])


def test_refined_reciprocal_is_the_nearest_float_for_every_mantissa(hip):
    """Markstein: the one-correction quotient q = fma(fma(-b, a rb, a), rb, a rb) is correctly rounded for EVERY numerator a if rb is
    the float nearest to 1 / b.  Exhaustive over the 2^23 mantissas of b (the reciprocal's error does not depend on the exponent; two
    exponents and a sign are checked anyway): the projection's refined reciprocal (v_rcp_f32 + one Newton step) against the exactly
    decided nearest float - 1 - b q is exact in float64 for float32 b and q, so q is nearest iff |1 - b q| is not larger than for
    either neighbour.  This turns the sampled "no mismatch seen" of rounds 2-3 into a proof for the whole operand space."""
    ctx, _ = hip
    m = np.arange(1 << 23, dtype=np.uint32)
    for expo, sign in ((127, 0), (121, 0), (130, 1)):
        b = ((np.uint32(sign) << np.uint32(31)) | (np.uint32(expo) << np.uint32(23)) | m).view(np.float32)
        rb, ieee = ctx.debug_divide(None, b)
        bd, q = b.astype(np.float64), rb.astype(np.float64)
        e0 = np.abs(1.0 - bd * q)
        up = (rb.view(np.uint32) + np.uint32(1)).view(np.float32).astype(np.float64)
        dn = (rb.view(np.uint32) - np.uint32(1)).view(np.float32).astype(np.float64)
        assert np.all(e0 <= np.abs(1.0 - bd * up)) and np.all(e0 <= np.abs(1.0 - bd * dn)), (expo, sign)
        assert np.array_equal(rb.view(np.uint32), ieee.view(np.uint32))          # = the IEEE division 1 / b, bit for bit


# ------------------------------------------------------------------------------------------------
# variant V3: the reference's numba.cuda kernel semantics (parity unpinned: oracle restatement only)
# ------------------------------------------------------------------------------------------------
def test_variant_v3_dense_and_increments_vs_oracle(hip, frames, pyramids):
    """cuda/residuals_kernel.py:46-109 restated with numba's arithmetic types (float64 geometry / Jacobian terms,
    float32 stores) in both the oracle and the kernel: validity and J are bit-identical, r to float32 rounding."""
    ctx, cabi = hip
    for lvl, xi in ((2, [0.002, -0.007, -0.011, 0.005, 0.0065, 0.0034]), (1, [0, 0, 0, 0, 0, 0]), (3, [0.01, 0.004, -0.02, -0.01, 0.008, 0.015])):
        T = O.Pose.from_xi(np.array(xi, dtype=np.float64)).matrix()
        r, J, valid, uv, (h, w) = ctx.residuals_jacobian(lvl, 0, 1, T, variant=cabi.VARIANT_V3)
        ro, Jo, mask_o, valid_o = O.residuals_and_jacobian(_level(frames, pyramids, 0, lvl), T, "v3")
        assert np.array_equal(valid.reshape(h, w), valid_o)
        assert np.array_equal(J[valid], Jo)                                # float64 expressions, float32 store: exact
        np.testing.assert_allclose(r[valid], ro.reshape(-1), rtol=0, atol=1e-4)
        rec = ctx.normal_equations(lvl, [0], [1], T[None], cabi.VARIANT_V3)[0]
        H, b, swr2, n, _ = cabi.unpack_record(rec)
        Ho, bo, so, no = O.normal_equations(ro, Jo)
        assert n == no and _rel(H, Ho) < 1e-6 and _rel(_solve(H, b), O.solve_increment(Ho, bo)) < 1e-5


def test_variant_v3_full_solve_vs_oracle(frames):
    import dvo_amd
    cam = dvo_amd.RGBDCameraModel(frames["K"], frames["depth_scale"])
    est = O.gauss_newton(frames["gray"][0], frames["depth"][0], frames["gray"][1], frames["K"], frames["depth_scale"],
                         levels=4, max_iterations=10, variant="v3")
    for solver in ("host", "device"):
        dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=10,
                              reference_variant="cuda", solver=solver)
        dvo.step_gray(frames["gray"][0], frames["depth"][0])
        tf = dvo.step_gray(frames["gray"][1], frames["depth"][1])
        np.testing.assert_allclose(tf.log().reshape(6), est.log(), rtol=0, atol=2e-6)
    dvo._setup(0)
    r, J, mask = dvo.compute_residuals_and_jacobian(dvo_amd.Se3.identity(), level=0)
    assert mask.sum() == r.shape[0]          # this variant's mask is depth-valid AND in bounds


# ------------------------------------------------------------------------------------------------
# variant GN: build-defined production mode: V1 with the gradient normalised and J_w at the warped point
# (parity unpinned: no reference counterpart; checked against the oracle's float64 restatement of the same definition)
# ------------------------------------------------------------------------------------------------
def test_variant_gn_dense_and_normal_equations_vs_oracle(hip, frames, pyramids):
    ctx, cabi = hip
    for lvl, xi in ((0, [0.002, -0.007, -0.011, 0.005, 0.0065, 0.0034]), (2, [0, 0, 0, 0, 0, 0]), (3, [0.01, 0.004, -0.02, -0.01, 0.008, 0.015])):
        T = O.Pose.from_xi(np.array(xi, dtype=np.float64)).matrix()
        r, J, valid, uv, (h, w) = ctx.residuals_jacobian(lvl, 0, 1, T, variant=cabi.VARIANT_GN)
        ro, Jo, mask_o, valid_o = O.residuals_and_jacobian(_level(frames, pyramids, 0, lvl), T, "gn")
        r1, J1, valid1, _, _ = ctx.residuals_jacobian(lvl, 0, 1, T, variant=cabi.VARIANT_V1)
        assert np.array_equal(valid.reshape(h, w), valid_o) and np.array_equal(valid, valid1)
        assert np.array_equal(r[valid], r1[valid1])                              # V1's residual, bit for bit
        scale = np.abs(Jo).max(axis=0)
        assert np.all(np.abs(J[valid] - Jo).max(axis=0) <= 3e-6 * scale), np.abs(J[valid] - Jo).max(axis=0) / scale
        for wm, wp in ((cabi.WEIGHT_NONE, None), (cabi.WEIGHT_HUBER, [10.0])):
            rec = ctx.normal_equations(lvl, [0], [1], T[None], cabi.VARIANT_GN, wm, wp)[0]
            H, b, swr2, n, _ = cabi.unpack_record(rec)
            wts = None if wp is None else O.huber_weights(ro, wp[0])
            Ho, bo, so, no = O.normal_equations(ro, Jo, wts)
            assert n == no and _rel(H, Ho) < 2e-6 and _rel(_solve(H, b), O.solve_increment(Ho, bo)) < 1e-5


def test_variant_gn_takes_full_gauss_newton_steps(frames):
    """The point of the mode: the reference's V1 takes 1/8 of a Gauss-Newton step per iteration (unnormalised Sobel, quirk Q1), so
    at 10 iterations per level it is still short of the minimum; GN gets at least as close in 3.  Device and host solve against
    the oracle's loop with the same definition on the fixture pair, then against exact ground truth on a rendered pair."""
    import dvo_amd
    from dvo_amd import synthetic
    cam = dvo_amd.RGBDCameraModel(frames["K"], frames["depth_scale"])
    est = O.gauss_newton(frames["gray"][0], frames["depth"][0], frames["gray"][1], frames["K"], frames["depth_scale"],
                         levels=4, max_iterations=3, variant="gn")
    for solver in ("device", "host"):
        dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, solver=solver, reference_variant="gn", max_iterations=3)
        dvo.step_gray(frames["gray"][0], frames["depth"][0])
        tf = dvo.step_gray(frames["gray"][1], frames["depth"][1])
        np.testing.assert_allclose(tf.log().reshape(6), est.log(), rtol=0, atol=5e-6)

    st = synthetic.make_stream(4, 640, 480, seed=11, device="cuda")
    scam = dvo_amd.RGBDCameraModel(st["K"], st["depth_scale"])
    errs = {}
    for name, kw in (("gn3", dict(reference_variant="gn", max_iterations=3)), ("v1_10", dict(max_iterations=10)),
                     ("v1_3", dict(max_iterations=3))):
        dvo = dvo_amd.get_dvo("robust-dvo", scam, dvo_amd.Se3.identity(), levels=4, solver="device", **kw)
        e = []
        for k in range(4):
            tf = dvo.step_gray(st["gray"][k], st["depth"][k])
            if k:
                rel_gt = np.linalg.inv(st["T_wc"][k]) @ st["T_wc"][k - 1]
                e.append(float(np.linalg.norm(tf.exp()[:3, 3] - rel_gt[:3, 3])))
        errs[name] = float(np.mean(e))
    assert errs["gn3"] < errs["v1_3"] and errs["gn3"] <= 1.05 * errs["v1_10"], errs


def test_degenerate_poses_give_empty_systems(hip):
    """NaN / far-away / behind-the-camera poses: every pixel fails the in-bounds test, nothing is read out of bounds,
    the record is exactly zero and the device loop stops after one iteration without touching the pose."""
    ctx, cabi = hip
    I = np.eye(4, dtype=np.float32)
    Tnan = I.copy(); Tnan[0, 3] = np.nan
    Tfar = I.copy(); Tfar[0, 3] = 1e6
    Tbehind = I.copy(); Tbehind[2, 3] = -100.0
    Tflip = np.diag([1, -1, -1, 1]).astype(np.float32)          # looks backwards: z' < 0, coordinates mirror
    for T in (Tnan, Tfar):
        rec = ctx.normal_equations(1, [0], [1], T[None])[0]
        assert (rec == 0).all()
    for T in (Tbehind, Tflip):                                   # the reference does not test z' > 0 either (quirk kept)
        rec = ctx.normal_equations(1, [0], [1], T[None])[0]
        assert np.isfinite(rec).all()
    res = ctx.solve_batch([0], [1], levels=4, max_iterations=10, T_init=Tfar[None])
    assert res["iters"][0] == 4                                  # one evaluated iteration per level, each stops at once
    np.testing.assert_array_equal(res["T"][0], Tfar[:3])


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("DVO_FUZZ_SEEDS", "6")))))      # DVO_FUZZ_SEEDS=200 for a longer soak
def test_fuzz_random_geometry_vs_oracle(seed):
    """Random image sizes (odd, not multiples of 4 or 64), intrinsics with and without skew, several depth scales, random
    depth-valid fractions (0.4 .. 1.0), random poses and weights: valid-point lists, validity and warped coordinates
    bit-exact, normal equations and increments within tolerance, against the oracle."""
    import dvo_amd
    from dvo_amd import _cabi
    rng = np.random.default_rng(100 + seed)
    h = int(rng.integers(17, 150))
    w = int(rng.integers(17, 210))
    levels = int(rng.integers(1, 4))
    while (min(h, w) >> (levels - 1)) < 4:
        levels -= 1
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    tex = 120 + 70 * np.sin(xx * rng.uniform(0.1, 0.5)) * np.cos(yy * rng.uniform(0.1, 0.5)) + rng.normal(0, 6, (h, w))
    g1 = np.clip(tex, 0, 255).astype(np.uint8)
    g2 = np.clip(np.roll(tex, int(rng.integers(-2, 3)), axis=1) + rng.normal(0, 3, (h, w)), 0, 255).astype(np.uint8)
    depth = rng.integers(2000, 30000, size=(h, w)).astype(np.uint16)
    depth[rng.random((h, w)) < rng.uniform(0.0, 0.6)] = 0
    depth_scale = [0.0002, 1.0 / 5000.0, 0.001, 1.0 / 3.0e4, float(rng.uniform(1e-4, 1e-3)), 1.0][seed % 6]
    skew = 0.0 if seed % 2 == 0 else float(rng.uniform(-3, 3))
    K = np.array([[rng.uniform(0.7, 1.4) * w, skew, w / 2 + rng.uniform(-5, 5)],
                  [0, rng.uniform(0.7, 1.4) * w, h / 2 + rng.uniform(-5, 5)], [0, 0, 1]], dtype=np.float32)
    with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=levels, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], depth_scale)
        ctx.upload_frame(0, g1, depth, levels)
        ctx.upload_frame(1, g2, depth, levels)
        gp, dp, gc = O.build_pyramid(g1, levels), O.build_pyramid(depth, levels), O.build_pyramid(g2, levels)
        for lvl in range(levels):
            xi = rng.normal(0, 1, 6) * np.array([0.02, 0.02, 0.02, 0.01, 0.01, 0.01])
            T = O.Pose.from_xi(xi).matrix()
            lv = O.PairLevel(gp[lvl], dp[lvl], gc[lvl], O.level_intrinsics(K, lvl), depth_scale)
            assert np.array_equal(ctx.level_intrinsics(lvl), lv.KL)
            pts = ctx.download_points(0, lvl)                                 # valid-point list: bit exact, raster order
            ys, xs = np.nonzero(dp[lvl])
            assert np.array_equal(pts["x"], xs) and np.array_equal(pts["y"], ys)
            assert np.array_equal(pts["intensity"], gp[lvl][ys, xs])
            assert np.array_equal(pts["z"], (dp[lvl][ys, xs].astype(np.float64) * depth_scale).astype(np.float32))
            for variant, vname in ((_cabi.VARIANT_V1, "v1"), (_cabi.VARIANT_V2, "v2"), (_cabi.VARIANT_V3, "v3"), (_cabi.VARIANT_GN, "gn")):
                if vname == "v3" and skew != 0.0:
                    continue                                   # the numba.cuda kernel has no skew term at all
                r, J, valid, uv, (hh, ww) = ctx.residuals_jacobian(lvl, 0, 1, T, variant=variant)
                ro, Jo, _, valid_o = O.residuals_and_jacobian(lv, T, vname)
                assert np.array_equal(valid.reshape(hh, ww), valid_o), (vname, lvl)
                if vname == "v1":
                    x, y, z, m = O.deproject(lv.depth_prev, lv.KL, lv.depth_scale)
                    uo, vo = O.project(lv.KL, *O.transform_points(T, x, y, z))
                    assert np.array_equal(uv[m.reshape(-1), 0], uo, equal_nan=True)
                    assert np.array_equal(uv[m.reshape(-1), 1], vo, equal_nan=True)
                if ro.shape[0] == 0:
                    continue
                np.testing.assert_allclose(r[valid], ro.reshape(-1), rtol=0, atol=2e-4)
                scale = np.maximum(np.abs(Jo).max(axis=0), 1e-6)
                np.testing.assert_allclose(J[valid] / scale, Jo / scale, rtol=0, atol=4e-6)
                w_mode, w_par, w_or = _cabi.WEIGHT_NONE, None, None
                if seed % 3 == 1:
                    k = 1.345 * float(np.sqrt(np.mean(ro.astype(np.float64) ** 2)))
                    w_mode, w_par, w_or = _cabi.WEIGHT_HUBER, [k], O.huber_weights(ro, float(np.float32(k)))
                rec = ctx.normal_equations(lvl, [0], [1], T[None], variant, w_mode, w_par)[0]
                H, b, swr2, n, sr2 = _cabi.unpack_record(rec)
                Ho, bo, so, no = O.normal_equations(ro, Jo, w_or)
                assert n == no
                assert _rel(H, Ho) < 2e-6 and abs(swr2 - so) <= 2e-6 * max(so, 1e-9)
                if no > 200 and np.linalg.cond(Ho) < 1e8:
                    assert _rel(_solve(H, b), O.solve_increment(Ho, bo)) < 1e-4 * max(1.0, np.linalg.cond(Ho) / 1e4)


def test_fused_level0_ingest_equals_separate_kernels(frames, monkeypatch):
    """Images with width % 8 == 0 and even height take the fused level-0 ingest kernel (copy + Sobel/pack + level-1
    median + valid-pixel counts in one pass); DVO_INGEST_UNFUSED forces the separate kernels every other geometry
    uses.  Pyramids, gradients and normal-equation records (i.e. the valid-point lists) must be bit-identical, for
    the single-frame upload (level 0 copied into the slot by the host call) and the batched one."""
    import dvo_amd
    T = O.Pose.from_xi(np.array([0.002, -0.007, -0.011, 0.005, 0.0065, 0.0034])).matrix()[None]
    got = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("DVO_INGEST_UNFUSED", "1")
        else:
            monkeypatch.delenv("DVO_INGEST_UNFUSED", raising=False)
        with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=4, max_pairs=1) as ctx:
            K = frames["K"]
            ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], frames["depth_scale"])
            ctx.upload_frame(0, frames["gray"][0], frames["depth"][0], 4)
            ctx.upload_frame(1, frames["gray"][1], frames["depth"][1], 4)
            ctx.upload_frames([2, 3], frames["gray"][2:4], frames["depth"][2:4], 4)
            ctx.synchronize()
            levels = [ctx.download_level(s, l) for s in range(4) for l in range(4)]
            recs = [ctx.normal_equations(l, [0], [1], T) for l in range(4)] + [ctx.normal_equations(l, [2], [3], T) for l in range(4)]
            got.append((levels, recs))
    for a, b in zip(got[0][0], got[1][0]):
        for k in ("gray", "depth", "gx", "gy"):
            assert np.array_equal(a[k], b[k]), k
    for a, b in zip(got[0][1], got[1][1]):
        assert np.array_equal(a, b)


def test_frames_uploaded_before_the_camera_must_be_uploaded_again(frames):
    """The valid-point lists built at ingest hold z = depth * depth_scale: a frame uploaded before dvo_set_camera (or
    under another depth_scale) still has its pyramid, but the alignment entry points refuse it (BAD_STATE)."""
    import dvo_amd
    from dvo_amd import _cabi
    T = np.eye(4, dtype=np.float32)[None, :3]
    K = frames["K"]
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=2, n_slots=2, max_pairs=1) as ctx:
        ctx.upload_frame(0, frames["gray"][0], frames["depth"][0], 2)
        ctx.upload_frame(1, frames["gray"][1], frames["depth"][1], 2)
        assert ctx.download_level(0, 1)["gray"].shape == (240, 320)
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], frames["depth_scale"])
        with pytest.raises(_cabi.DVOHipError) as e:
            ctx.normal_equations(0, [0], [1], T)
        assert e.value.code == -5
        ctx.upload_frame(0, frames["gray"][0], frames["depth"][0], 2)
        ctx.upload_frame(1, frames["gray"][1], frames["depth"][1], 2)
        ref = ctx.normal_equations(0, [0], [1], T)
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], 2.0 * frames["depth_scale"])
        with pytest.raises(_cabi.DVOHipError):
            ctx.normal_equations(0, [0], [1], T)
        ctx.set_camera(K[0, 0] * 1.01, K[1, 1], K[0, 2], K[1, 2], K[0, 1], 2.0 * frames["depth_scale"])   # intrinsics only: lists stay valid
        ctx.upload_frame(0, frames["gray"][0], frames["depth"][0], 2)
        assert ctx.normal_equations(0, [0], [1], T).shape == ref.shape


# ------------------------------------------------------------------------------------------------
# round-2 regressions: graph cache vs camera change, first-pair prior, step_gray guard, factory default
# ------------------------------------------------------------------------------------------------
def test_solve_graph_follows_a_camera_change(frames):
    """dvo_solve_batch replays a captured launch graph; the level intrinsics are kernel arguments inside it.  After
    dvo_set_camera (same image size, other K — e.g. TUM fr1 then fr2 on one ctx) the next solve must use the new
    intrinsics: bit-identical to a fresh context, and different from the solve under the old camera."""
    import dvo_amd
    K, sc = frames["K"], frames["depth_scale"]

    def solve(ctx):
        return ctx.solve_batch([0, 1], [1, 2], levels=4, max_iterations=6)

    def load(ctx, fx_scale, skew):
        ctx.set_camera(K[0, 0] * fx_scale, K[1, 1], K[0, 2], K[1, 2], skew, sc)
        ctx.upload_frames([0, 1, 2], frames["gray"][:3], frames["depth"][:3], 4)

    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=3, max_pairs=2) as ctx:
        load(ctx, 1.0, 0.0)
        a = solve(ctx)
        a2 = solve(ctx)                                   # replayed graph
        assert np.array_equal(a["xi"], a2["xi"])
        load(ctx, 1.01, 0.0)
        b = solve(ctx)
        load(ctx, 1.01, 0.3)                              # skew switches the kernel instantiation inside the graph
        c = solve(ctx)
        ctx.set_camera(K[0, 0] * 1.01, K[1, 1], K[0, 2], K[1, 2], 0.0, sc)     # intrinsics only: no re-upload needed
        b2 = solve(ctx)
    fresh = []
    for fx_scale, skew in ((1.01, 0.0), (1.01, 0.3)):
        with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=3, max_pairs=2) as ctx:
            load(ctx, fx_scale, skew)
            fresh.append(solve(ctx))
    assert np.array_equal(b["xi"], fresh[0]["xi"]) and np.array_equal(b["T"], fresh[0]["T"])
    assert np.array_equal(c["xi"], fresh[1]["xi"])
    assert np.array_equal(b2["xi"], fresh[0]["xi"])
    assert np.abs(a["xi"] - b["xi"]).max() > 1e-6 and np.abs(b["xi"] - c["xi"]).max() > 1e-6


def test_prior_without_previous_transform_is_no_prior(frames):
    """First pair of a sequence: `old is None`, the reference applies no prior (base_robust_dvo.py:154,:190).
    dvo_set_prior(sigma, NULL) through the C ABI must therefore solve exactly like sigma off."""
    import dvo_amd
    K = frames["K"]
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], frames["depth_scale"])
        ctx.upload_frames([0, 1], frames["gray"][:2], frames["depth"][:2], 4)
        ctx.set_prior(None)
        a = ctx.solve_batch([0], [1], levels=4, max_iterations=8)
        ctx.set_prior(1e-13, None)
        b = ctx.solve_batch([0], [1], levels=4, max_iterations=8)
        qt = np.array([[1.0, 0, 0, 0, 0, 0, 0]])
        ctx.set_prior(1e-13, qt)                          # an explicit identity `last` DOES damp the step
        c = ctx.solve_batch([0], [1], levels=4, max_iterations=8)
    assert np.array_equal(a["xi"], b["xi"]) and np.array_equal(a["iters"], b["iters"])
    assert np.abs(a["xi"] - c["xi"]).max() > 1e-5


@pytest.mark.parametrize("entry", ["step", "step_gray"])
def test_step_keeps_state_when_no_transform_is_found(frames, entry, monkeypatch):
    """core/base_dense_visual_odometry.py:76-87: a `None` transform leaves pose and previous frame untouched and is
    returned to the caller (both `step` and `step_gray`)."""
    dvo, dvo_amd = _make_dvo(frames, levels=2, max_iterations=3, solver="device")

    def feed(k):
        if entry == "step":
            return dvo.step(np.repeat(frames["gray"][k][..., None], 3, axis=2), frames["depth"][k].copy())
        return dvo.step_gray(frames["gray"][k], frames["depth"][k])

    feed(0)
    t1 = feed(1)
    pose1 = dvo.current_pose.exp().copy()
    monkeypatch.setattr(dvo, "_gauss_newton", lambda init_guess: None)
    assert feed(2) is None
    assert np.array_equal(dvo.current_pose.exp(), pose1)
    monkeypatch.undo()
    t3 = feed(2)                                          # still aligned against frame 1 (the failed frame was dropped)
    ref = O.gauss_newton(frames["gray"][1], frames["depth"][1], frames["gray"][2], frames["K"], frames["depth_scale"],
                         levels=2, max_iterations=3)
    np.testing.assert_allclose(t3.log().reshape(6), ref.log(), rtol=0, atol=2e-6)
    assert t1 is not None


def test_factory_accepts_the_reference_default_use_gpu_false(frames):
    """core/robust_dense_visual_odometry/__init__.py:5: the reference's factory defaults to use_gpu=False; a reference
    config carrying it must load (the HIP backend reproduces the CPU backend's semantics)."""
    import dvo_amd
    cam = dvo_amd.RGBDCameraModel(frames["K"], frames["depth_scale"])
    a = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=1, use_gpu=False, height=48, width=64)
    b = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=1, height=48, width=64)
    assert type(a) is type(b) is dvo_amd.RobustDVOHIP
    with pytest.raises(ValueError):
        dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=1, backend="cuda")


# ------------------------------------------------------------------------------------------------
# BASELINE config 3 at full size against the oracle (not only through properties)
# ------------------------------------------------------------------------------------------------
def test_config3_1280x960_normal_equations_vs_oracle():
    """1280x960, 5-level pyramid: the fused reduction at the finest and the coarsest level, unweighted and Huber,
    against oracle.residuals_and_jacobian + normal_equations on the same seeded synthetic pair (N exact, H / b /
    increment tolerances as at VGA)."""
    import dvo_amd
    from dvo_amd import synthetic, _cabi
    st = synthetic.make_stream(2, 1280, 960, seed=5, device="cuda")
    K, sc = np.asarray(st["K"], dtype=np.float32), st["depth_scale"]
    g, d = st["gray"], st["depth"]
    gp, dp, gc = O.build_pyramid(g[0], 5), O.build_pyramid(d[0], 5), O.build_pyramid(g[1], 5)
    T = O.Pose.from_xi(np.array([0.004, -0.003, 0.006, 0.003, -0.002, 0.0015])).matrix()
    with dvo_amd.HipContext(max_w=1280, max_h=960, max_levels=5, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], sc)
        ctx.upload_frames([0, 1], g[:2], d[:2], 5)
        for lvl in (0, 4):
            lv = O.PairLevel(gp[lvl], dp[lvl], gc[lvl], O.level_intrinsics(K, lvl), sc)
            r, J, _, _ = O.residuals_and_jacobian(lv, T)
            k = 1.345 * float(np.sqrt(np.mean(r.astype(np.float64) ** 2)))
            for mode, par, w in ((_cabi.WEIGHT_NONE, None, None),
                                 (_cabi.WEIGHT_HUBER, [k], O.huber_weights(r, float(np.float32(k))))):
                rec = ctx.normal_equations(lvl, [0], [1], T[None], _cabi.VARIANT_V1, mode, par)[0]
                H, b, swr2, n, _ = _cabi.unpack_record(rec)
                Ho, bo, so, no = O.normal_equations(r, J, w)
                assert n == no and n > (300_000 if lvl == 0 else 1_000)
                assert _rel(H, Ho) < 1e-6 and _rel(b, bo) < 2e-6 and abs(swr2 - so) < 2e-6 * so
                assert _rel(_solve(H, b), O.solve_increment(Ho, bo)) < 1e-5
