"""Pins the CPU oracle (oracle/dvo_oracle.py) to the reference.

Golden vectors in tests/golden/*.npz were produced by oracle/refrun/make_golden.py, which executes
the reference's own source files (GN loop, CPU backend, camera, Lie classes) in the build
container.  Also restates the reference's own known-answer tests for this path:
  tests/dense_visual_odometry/core/robust_dense_visual_odometry/test_cpu_robust_dense_visual_odometry.py:20-73
  tests/dense_visual_odometry/utils/test_jacobian.py:16-38, :49-73
  tests/dense_visual_odometry/test_camera_model.py:110-180
"""
import numpy as np
import pytest

import dvo_oracle as O


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


# ---------------------------------------------------------------------------------------------
# inputs of the hot path: pyramids, gradients, intrinsics
# ---------------------------------------------------------------------------------------------
def test_pyramid_gradients_intrinsics_exact(frames, golden_dir):
    py = np.load(golden_dir / "pyramid_f1f2.npz")
    gp = O.build_pyramid(frames["gray"][0], 4)
    dp = O.build_pyramid(frames["depth"][0], 4)
    gc = O.build_pyramid(frames["gray"][1], 4)
    for l in range(1, 4):
        assert gp[l].shape == (480 >> l, 640 >> l)
        assert np.array_equal(gp[l], py["gray1_l%d" % l])
        assert np.array_equal(dp[l], py["depth1_l%d" % l])
        assert np.array_equal(gc[l], py["gray2_l%d" % l])
    for l in range(4):
        gx, gy = O.sobel3(gc[l])
        assert np.array_equal(gx, py["gradx2_l%d" % l].astype(np.float32))
        assert np.array_equal(gy, py["grady2_l%d" % l].astype(np.float32))
    for l in range(5):
        assert np.array_equal(O.level_intrinsics(frames["K"], l), frames["level_intrinsics"][l])


def test_pyramid_odd_sizes():
    # ImagePyramidGPU uses ceil(h/2) (image_pyramid.py:83-84); [::2, ::2] gives the same
    img = (np.arange(7 * 9) % 251).astype(np.uint8).reshape(7, 9)
    p = O.build_pyramid(img, 3)
    assert p[1].shape == (4, 5) and p[2].shape == (2, 3)
    with pytest.raises(ValueError):
        O.build_pyramid(img, -1)


def test_sobel_known_answer():
    # reference tests/dense_visual_odometry/utils/test_jacobian.py:49-73
    image = np.zeros((5, 5), dtype=np.float32)
    image[2, :] = 1.0
    gx, gy = O.sobel3(image)
    expected = np.zeros((5, 5), dtype=np.float32)
    expected[1, :] = 4.0
    expected[3, :] = -4.0
    np.testing.assert_almost_equal(gx, np.zeros_like(image))
    np.testing.assert_almost_equal(gy, expected)


def test_warp_jacobian_known_answer():
    # reference tests/dense_visual_odometry/utils/test_jacobian.py:16-38
    pc = np.array([[1, 4], [2, 5], [3, 6], [1, 1]], dtype=np.float32)
    K = np.zeros((3, 4), dtype=np.float32)
    K[:3, :3] = np.eye(3)
    res = O.warp_jacobian(pc[0], pc[1], pc[2], K)
    expected = np.array([
        [[1 / 3, 0, -(1 / 9), -(2 / 9), 10 / 9, -(2 / 3)], [0, 1 / 3, -(2 / 9), -(13 / 9), 2 / 9, 1 / 3]],
        [[1 / 6, 0, -(1 / 9), -(5 / 9), 13 / 9, -(5 / 6)], [0, 1 / 6, -(5 / 36), -(61 / 36), 5 / 9, 2 / 3]]],
        dtype=np.float32)
    np.testing.assert_allclose(res, expected, atol=1e-5)


def test_deproject_project_known_answer():
    # reference tests/dense_visual_odometry/test_camera_model.py:110-180 (K = I, 10x10 / 5x5)
    K = np.zeros((3, 4), dtype=np.float32)
    K[:3, :3] = np.eye(3)
    depth = np.zeros((10, 10), dtype=np.uint16)
    depth[2:5, 3:7] = 2
    x, y, z, mask = O.deproject(depth, K, 1.0)
    vv, uu = np.nonzero(depth)
    np.testing.assert_array_equal(z, np.full(12, 2, dtype=np.float32))
    np.testing.assert_array_equal(x, (uu * 2).astype(np.float32))
    np.testing.assert_array_equal(y, (vv * 2).astype(np.float32))
    assert np.array_equal(mask, depth != 0)
    u, v = O.project(K, x, y, z)
    np.testing.assert_array_equal(u, uu.astype(np.float32))
    np.testing.assert_array_equal(v, vv.astype(np.float32))


def test_tiny10_reference_known_answer(golden_dir):
    # reference test_cpu_robust_dense_visual_odometry.py:20-44: 100 residuals, all zero
    t = np.load(golden_dir / "tiny10.npz")
    K = np.zeros((3, 4), dtype=np.float32)
    K[:3, :3] = np.eye(3)
    lv = O.PairLevel(t["gray"], t["depth"], t["gray"], K, 1.0)
    r, J, mask, valid = O.residuals_and_jacobian(lv, np.eye(4, dtype=np.float32))
    assert r.shape == (100, 1)
    np.testing.assert_almost_equal(r, np.zeros((100, 1), dtype=np.float32))
    assert np.array_equal(J, t["J"]) and np.array_equal(mask, t["mask"]) and valid.all()


def test_residuals_with_ground_truth_pose(frames):
    # reference test_cpu_robust_dense_visual_odometry.py:46-73 (pairs (4,5), (5,6); 1-based)
    for idx in (4, 5):
        T0, T1 = frames["gt_T"][idx - 1], frames["gt_T"][idx]
        Trel = (np.linalg.inv(T1) @ T0).astype(np.float32)
        lv = O.PairLevel(frames["gray"][idx - 1], frames["depth"][idx - 1], frames["gray"][idx],
                         O.level_intrinsics(frames["K"], 0), frames["depth_scale"])
        r, _, _, _ = O.residuals_and_jacobian(lv, Trel)
        assert np.isclose(r.mean(), 0.0, atol=5.0)
        assert r.std() < 20


# ---------------------------------------------------------------------------------------------
# dense r / J / mask against the reference run
# ---------------------------------------------------------------------------------------------
def test_dense_residuals_jacobian_bit_exact(frames, golden_dir):
    d = np.load(golden_dir / "dense_rj.npz")
    gp = O.build_pyramid(frames["gray"][0], 4)
    dp = O.build_pyramid(frames["depth"][0], 4)
    gc = O.build_pyramid(frames["gray"][1], 4)
    for lvl in (3, 2):
        for tag in ("id", "xi"):
            pre = "l%d_%s_" % (lvl, tag)
            lv = O.PairLevel(gp[lvl], dp[lvl], gc[lvl], O.level_intrinsics(frames["K"], lvl), frames["depth_scale"])
            r, J, mask, valid = O.residuals_and_jacobian(lv, d[pre + "T"])
            assert np.array_equal(mask, d[pre + "mask"])
            assert r.shape == d[pre + "r"].shape and int(valid.sum()) == r.shape[0]
            # bit for bit against the reference run (which evaluated J_w with the generator's vectorised stand-in of
            # utils/jacobian.py:7-44; the reference's interpreted per-point loop differs from that stand-in by float32
            # rounding only, 7.6e-6 absolute: meta.json "vectorised_loop_check")
            assert np.array_equal(r, d[pre + "r"])
            assert np.array_equal(J, d[pre + "J"])


# ---------------------------------------------------------------------------------------------
# per-iteration normal equations / increments against the reference run (teacher forced)
# ---------------------------------------------------------------------------------------------
def _check_iterations(frames, rec, pair_of, variant="v1", weighted=False, stride=1):
    cache = {}
    worst = 0.0
    for i in range(0, len(rec["level"]), stride):
        k = int(pair_of(i))
        lvl = int(rec["level"][i])
        if k not in cache:
            cache.clear()
            cache[k] = (O.build_pyramid(frames["gray"][k], 4), O.build_pyramid(frames["depth"][k], 4),
                        O.build_pyramid(frames["gray"][k + 1], 4), {})
        gp, dp, gc, lvs = cache[k]
        if lvl not in lvs:
            lvs[lvl] = O.PairLevel(gp[lvl], dp[lvl], gc[lvl], O.level_intrinsics(frames["K"], lvl), frames["depth_scale"])
        r, J, _, _ = O.residuals_and_jacobian(lvs[lvl], rec["T"][i], variant)
        w = O.student_t_weights(r)[0] if weighted else None
        H, b, s, n = O.normal_equations(r, J, w)
        assert n == int(rec["n"][i])
        inc = O.solve_increment(H, b)
        # tolerance stated by BASELINE.json north_star: 1e-5 relative on the increment
        assert _rel(inc, rec["inc64"][i]) < 1e-5
        assert _rel(H, rec["H64"][i]) < 1e-6 and _rel(b, rec["b64"][i]) < 1e-6
        assert abs(s - rec["sum_wr2"][i]) <= 1e-6 * abs(rec["sum_wr2"][i])
        # the reference's own float32 BLAS path is only reproducible to ~1e-5..1e-4
        assert _rel(inc, rec["inc32"][i]) < 5e-4
        worst = max(worst, _rel(inc, rec["inc64"][i]))
    return worst


def test_increments_nine_pairs_four_levels(frames, golden_dir):
    rec = np.load(golden_dir / "pairs_l4.npz")
    worst = _check_iterations(frames, rec, lambda i: rec["pair"][i])
    assert worst < 1e-8


def test_increments_single_level_trace(frames, golden_dir):
    rec = np.load(golden_dir / "single_l1.npz")
    _check_iterations(frames, rec, lambda i: 0, stride=4)


def test_increments_student_t_reference_weights(frames, golden_dir):
    rec = np.load(golden_dir / "weighted_l4.npz")
    _check_iterations(frames, rec, lambda i: 0, weighted=True, stride=3)
    # quirk Q3: the "t-distribution" weights are ~1.2 for every pixel
    assert rec["w_min"].min() > 1.19 and rec["w_max"].max() <= 1.2 + 1e-6


def test_increments_variant_v2(frames, golden_dir):
    rec = np.load(golden_dir / "approx_l4.npz")
    _check_iterations(frames, rec, lambda i: 0, variant="v2", stride=3)


# ---------------------------------------------------------------------------------------------
# full Gauss-Newton driver
# ---------------------------------------------------------------------------------------------
def test_gauss_newton_single_level_to_convergence(frames, golden_dir):
    g = np.load(golden_dir / "single_l1.npz")
    rec = []
    est = O.gauss_newton(frames["gray"][0], frames["depth"][0], frames["gray"][1], frames["K"],
                         frames["depth_scale"], levels=1, record=rec)
    assert len(rec) == len(g["level"]) == 54
    np.testing.assert_allclose(est.log(), g["final_xi"], rtol=0, atol=2e-7)


@pytest.mark.parametrize("k", [0, 4, 7])
def test_gauss_newton_four_levels(frames, golden_dir, k):
    g = np.load(golden_dir / "pairs_l4.npz")
    rec = []
    est = O.gauss_newton(frames["gray"][k], frames["depth"][k], frames["gray"][k + 1], frames["K"],
                         frames["depth_scale"], levels=4, max_iterations=10, record=rec)
    assert len(rec) == int((g["pair"] == k).sum())          # pair 7 stops early at one level (36)
    np.testing.assert_allclose(est.log(), g["final_xi"][k], rtol=0, atol=2e-7)
    np.testing.assert_allclose(est.matrix(), g["final_T"][k], rtol=0, atol=2e-7)


def test_gauss_newton_float32_mode_close_to_reference(frames, golden_dir):
    g = np.load(golden_dir / "pairs_l4.npz")
    est = O.gauss_newton(frames["gray"][0], frames["depth"][0], frames["gray"][1], frames["K"],
                         frames["depth_scale"], levels=4, max_iterations=10, accumulate="fp32")
    np.testing.assert_allclose(est.log(), g["final_xi"][0], rtol=0, atol=2e-6)


# ---------------------------------------------------------------------------------------------
# Lie groups against the reference classes
# ---------------------------------------------------------------------------------------------
def test_lie_vectors(golden_dir):
    g = np.load(golden_dir / "lie.npz")
    prev = O.Pose.identity()
    for i, xi in enumerate(g["xi"]):
        p = O.Pose.from_xi(xi.reshape(6, 1))
        np.testing.assert_allclose(p.matrix(), g["T"][i], atol=1e-6)
        np.testing.assert_allclose(p.log(), g["log"][i], atol=1e-6)
        np.testing.assert_allclose(p.quat, g["quat"][i], atol=1e-7)
        np.testing.assert_allclose(p.inverse().matrix(), g["inv_T"][i], atol=1e-6)
        np.testing.assert_allclose(p.mul(prev).matrix(), g["comp_T"][i], atol=1e-6)
        prev = p


def test_sequence_pose_chain(frames, golden_dir):
    """step() chaining, core/base_dense_visual_odometry.py:79: pose <- pose * T^-1."""
    g = np.load(golden_dir / "sequence.npz")
    pairs = np.load(golden_dir / "pairs_l4.npz")
    T0 = frames["gt_T"][0]
    pose = O.Pose(O.quat_from_matrix(T0[:3, :3]).astype(np.float64), T0[:3, 3].copy())
    np.testing.assert_allclose(pose.matrix(), g["pose_T"][0], atol=1e-6)
    for k in range(9):
        np.testing.assert_allclose(pairs["final_T"][k], g["tf_T"][k + 1], atol=1e-7)
        tf = O.Pose(pairs["final_quat"][k], pairs["final_tvec"][k])
        pose = pose.mul(tf.inverse())
        np.testing.assert_allclose(pose.matrix(), g["pose_T"][k + 1], atol=2e-6)


def test_gauss_newton_motion_prior_sequence(frames, golden_dir):
    """sigma prior (base_robust_dvo.py:190-194,:217-218): chained pairs, coupled through the last transform.
    The reference adds 1/sigma = 1e13 to a float32 H of ~1e14, so its own rounding is ~1e-6 on xi."""
    g = np.load(golden_dir / "sigma_l4.npz")
    last = O.Pose.identity()
    for k in range(3):
        rec = []
        est = O.gauss_newton(frames["gray"][k], frames["depth"][k], frames["gray"][k + 1], frames["K"],
                             frames["depth_scale"], levels=4, max_iterations=10, sigma=float(g["sigma"]),
                             last_transform=last, record=rec)
        assert len(rec) == int((g["pair"] == k).sum())            # [36, 22, 9]: the prior trips the stop rule early
        np.testing.assert_allclose(est.log(), g["final_xi"][k], rtol=0, atol=5e-6)
        last = est
    # the prior matters at this sigma: the unconstrained solution is far away
    free = np.load(golden_dir / "pairs_l4.npz")["final_xi"][0]
    assert np.abs(free - g["final_xi"][0]).max() > 1e-3


def test_variant_v3_strict_bounds_and_types(golden_dir, frames):
    """V3 = the reference's numba.cuda kernel (cuda/residuals_kernel.py:46-109): parity unpinned (cannot run here);
    these are the properties its source guarantees."""
    t = np.load(golden_dir / "tiny10.npz")
    K = np.zeros((3, 4), dtype=np.float32)
    K[:3, :3] = np.eye(3)
    lv = O.PairLevel(t["gray"], t["depth"], t["gray"], K, 1.0)
    r, J, mask, valid = O.residuals_and_jacobian(lv, np.eye(4, dtype=np.float32), "v3")
    # strict rule x1 < W, y1 < H (:30-31): the last row and column are invalid -> 9 x 9 pixels, mask == valid
    assert r.shape == (81, 1) and np.all(r == 0) and mask.sum() == 81 and not mask[-1].any() and not mask[:, -1].any()
    assert J.dtype == np.float32 and np.array_equal(mask, valid)
    # central differences are 1/16 of the unnormalised Sobel response on a linear ramp; J_w at the warped point
    ramp = np.tile((np.arange(64) * 3).astype(np.uint8), (48, 1))
    gx, gy = O.central_diff(ramp)
    sx, sy = O.sobel3(ramp)
    assert np.all(gx[:, 1:-1] == 3.0) and np.all(sx[:, 1:-1] == 24.0) and np.all(gy == 0)
    # on real data V3 and V1 agree on the residuals of the pixels both accept (same warp up to float32/float64 rounding)
    gp, dp, gc = (O.build_pyramid(frames["gray"][0], 3), O.build_pyramid(frames["depth"][0], 3),
                  O.build_pyramid(frames["gray"][1], 3))
    lv = O.PairLevel(gp[2], dp[2], gc[2], O.level_intrinsics(frames["K"], 2), frames["depth_scale"])
    T = O.Pose.from_xi(np.array([0.002, -0.007, -0.011, 0.005, 0.0065, 0.0034])).matrix()
    r3, J3, m3, v3 = O.residuals_and_jacobian(lv, T, "v3")
    r1, J1, m1, v1 = O.residuals_and_jacobian(lv, T, "v1")
    both = v1 & v3
    a = np.zeros(v1.shape, np.float32); a[v1] = r1.reshape(-1)
    b = np.zeros(v3.shape, np.float32); b[v3] = r3.reshape(-1)
    assert both.sum() > 0.95 * v1.sum() and np.abs(a[both] - b[both]).max() < 0.05


def test_variant_gn_is_v1_with_unit_gradients_and_the_warped_point(frames):
    """GN = the build's production mode (DVO_VARIANT_GN, no reference counterpart; parity unpinned).  What its definition
    guarantees: V1's residuals and validity; at the identity pose (warped point == un-warped point) exactly 1/8 of V1's
    Jacobian, hence 8 x V1's increment — V1 takes an eighth of a Gauss-Newton step (quirk Q1); away from the identity the rows
    differ by J_w's evaluation point only; and the loop gets closer to the minimum in 3 iterations per level than V1 does."""
    gp, dp, gc = (O.build_pyramid(frames["gray"][0], 4), O.build_pyramid(frames["depth"][0], 4),
                  O.build_pyramid(frames["gray"][1], 4))
    lv = O.PairLevel(gp[3], dp[3], gc[3], O.level_intrinsics(frames["K"], 3), frames["depth_scale"])
    I = np.eye(4, dtype=np.float32)
    r1, J1, m1, v1 = O.residuals_and_jacobian(lv, I, "v1")
    rg, Jg, mg, vg = O.residuals_and_jacobian(lv, I, "gn")
    assert np.array_equal(r1, rg) and np.array_equal(v1, vg) and np.array_equal(m1, mg)
    np.testing.assert_allclose(Jg * 8.0, J1, rtol=2e-6, atol=1e-3)
    H1, b1, _, _ = O.normal_equations(r1, J1)
    Hg, bg, _, _ = O.normal_equations(rg, Jg)
    np.testing.assert_allclose(O.solve_increment(Hg, bg), 8.0 * O.solve_increment(H1, b1), rtol=1e-5, atol=1e-9)
    T = O.Pose.from_xi(np.array([0.01, 0.004, -0.02, -0.01, 0.008, 0.015])).matrix()
    r1, J1, _, _ = O.residuals_and_jacobian(lv, T, "v1")
    rg, Jg, _, _ = O.residuals_and_jacobian(lv, T, "gn")
    assert np.array_equal(r1, rg) and not np.allclose(Jg * 8.0, J1, rtol=1e-4)
    # convergence on the two coarsest levels of the fixture pair (small enough for the CPU suite): final photometric error
    small = dict(gray_prev=gp[2], depth_prev=dp[2], gray_cur=gc[2], K=O.level_intrinsics(frames["K"], 2)[:3, :3],
                 depth_scale=frames["depth_scale"], levels=2)
    errs = {}
    for name, variant, iters in (("gn3", "gn", 3), ("v1_3", "v1", 3), ("v1_10", "v1", 10)):
        est = O.gauss_newton(small["gray_prev"], small["depth_prev"], small["gray_cur"], small["K"], small["depth_scale"],
                             levels=2, max_iterations=iters, variant=variant)
        lv0 = O.PairLevel(gp[2], dp[2], gc[2], O.level_intrinsics(frames["K"], 2), frames["depth_scale"])
        r, _, _, _ = O.residuals_and_jacobian(lv0, est.matrix(), "v1")
        errs[name] = float(np.mean(r.astype(np.float64) ** 2))
    # (real frames, coarse levels: the two end within a few percent of each other; 3 V1 iterations are nowhere near)
    assert errs["gn3"] < 0.6 * errs["v1_3"] and errs["gn3"] <= errs["v1_10"] * 1.05, errs


def test_cpu_baseline_worker_runs_the_oracle_on_a_sample(tmp_path, frames):
    """oracle/cpu_worker.py (bench.py's CPU-baseline leg, run as plain child processes): same poses as calling the oracle
    directly, and the command-line form prints the seconds of its range."""
    import subprocess
    import sys
    from pathlib import Path
    import cpu_worker
    path = tmp_path / "sample.npz"
    g, d = frames["gray"][:2, ::4, ::4], frames["depth"][:2, ::4, ::4]           # 160x120 keeps it quick
    K = O.level_intrinsics(frames["K"], 2)
    np.savez(path, gray=g, depth=d, K=K, depth_scale=frames["depth_scale"])
    secs, n = cpu_worker.run_pairs(str(path), 0, 1, 2, 2)
    assert n == 1 and secs > 0
    out = subprocess.run([sys.executable, str(Path(cpu_worker.__file__)), str(path), "0", "1", "2", "2"],
                         capture_output=True, text=True, check=True)
    assert float(out.stdout.strip().splitlines()[-1]) > 0


# ---------------------------------------------------------------------------------------------
# the one replaced reference function: interpolate_bilinear (cpu_robust...:202-254)
# ---------------------------------------------------------------------------------------------
def test_bilinear_replacement_equals_the_reference_function_where_it_is_defined(golden_dir):
    """make_golden.py keeps the reference's ORIGINAL interpolate_bilinear and runs it (plain Python) on every
    strictly-interior warped coordinate of a fixture level, for a uint8 image and a float32 gradient plane: the
    closed-boundary replacement used for all goldens must be bit-identical there (recorded in meta.json), and the
    oracle's `bilinear` must reproduce the original function's recorded outputs bit for bit."""
    import json
    meta = json.loads((golden_dir / "meta.json").read_text())["bilinear_replacement_check"]
    assert meta["u8_bit_exact"] is True and meta["f32_bit_exact"] is True
    assert meta["u8_max_abs_diff"] == 0.0 and meta["f32_max_abs_diff"] == 0.0
    assert meta["points_strictly_interior"] > 50_000
    g = np.load(golden_dir / "bilinear_ref.npz")
    uv = g["coords"]
    assert uv.dtype == np.float32 and uv.shape[0] >= 5000
    for name in ("u8", "f32"):
        got = O.bilinear(g["image_" + name], uv[:, 0], uv[:, 1])
        assert got.dtype == np.float32 and not np.isnan(got).any()
        assert np.array_equal(got.reshape(-1, 1), g["out_" + name])


# ---------------------------------------------------------------------------------------------
# the reference CLI's colour data flow (src/test_dvo.py:183 + core/base_dense_visual_odometry.py:58)
# ---------------------------------------------------------------------------------------------
def test_reference_cli_colour_data_flow(frames, golden_dir):
    """cli_compat.npz: fixture frames 1, 2 as the reference's CLI hands them to step() (RGB order into BGR2GRAY) and what its
    step() then estimates.  The oracle's gray conversion of the swapped-channel image and its solve on those planes reproduce
    both; the correct BGR order gives a (slightly) different gray plane — which is why the switch exists."""
    c = np.load(golden_dir / "cli_compat.npz")
    for k in (1, 2):
        assert np.array_equal(O.bgr2gray(c["rgb_%d" % k]), c["gray_cli_%d" % k])
    assert np.array_equal(O.bgr2gray(np.ascontiguousarray(c["rgb_1"][..., ::-1])), frames["gray"][0])      # true BGR order
    assert not np.array_equal(c["gray_cli_1"], frames["gray"][0])
    d = [c["depth_1"].copy(), c["depth_2"].copy()]
    for a in d:
        a[(a * frames["depth_scale"]) > 5.0] = 0                                   # base_dense_visual_odometry.py:59
    est = O.gauss_newton(c["gray_cli_1"], d[0], c["gray_cli_2"], frames["K"], frames["depth_scale"], levels=4, max_iterations=10)
    np.testing.assert_allclose(est.log(), c["tf_xi"], rtol=0, atol=2e-7)


def test_rank_rule_of_the_solve_vs_reference_runs(golden_dir):
    """a15: the reference solves with scipy lstsq / gelsy on a FLOAT32 H (base_robust_dvo.py:196-198), so LAPACK drops every direction
    whose singular value is below float32 eps of the largest.  rank_cases.npz holds what the reference's own step() solved on three
    synthetic pairs (oracle/refrun/make_golden.py: rank_cases): vertical stripes on a fronto-parallel plane (rank 5 under any
    rule), the same with a vertical gradient around two pixels (eigenvalue ratio 6.5e-9: rank 5 under the float32 rule, rank 6 under a
    float64 one - the increments differ by 4.5e-4 relative), and a weakly textured scene (full rank).  The oracle's solve applies the
    reference's rule to float64 data and reproduces the reference's float32 increments to their own rounding."""
    g = np.load(golden_dir / "rank_cases.npz")
    for name, dropped in (("stripes", True), ("window", True), ("textured", False)):
        for j in range(3):
            H, b, ref = g[name + "_H64"][j], g[name + "_b64"][j], g[name + "_inc32"][j].astype(np.float64)
            inc = O.solve_increment(H, b)
            assert _rel(inc, ref) < 5e-4, (name, j, _rel(inc, ref))                 # float32 normal equations vs float64 ones
            full = np.linalg.solve(H + 0.0, b) if np.linalg.matrix_rank(H) == 6 else None
            if dropped:
                assert abs(inc[1]) < 1e-8 and abs(ref[1]) < 1e-8                    # no step along the direction the data do not constrain
            if name == "window":
                assert _rel(full, ref) > 2e-4 and abs(full[1]) > 5e-7               # a float64 rank rule takes that step: not the reference
            if name == "textured":
                assert _rel(full, inc) < 1e-9
    # and the whole loop of the oracle on the window pair ends where the reference run did
    est = O.gauss_newton(g["window_gray1"], g["depth"], g["window_gray2"], g["intrinsics"], float(g["depth_scale"]), levels=1, max_iterations=3)
    np.testing.assert_allclose(est.log(), g["window_final_xi"], rtol=0, atol=2e-6)

