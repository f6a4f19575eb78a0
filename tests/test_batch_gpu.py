"""BatchAligner / synthetic streams on the GPU: the batched device-side solve against ground truth and
against the oracle (BASELINE configs 2/3/5 stand-ins).  Run with -m gpu."""
import numpy as np
import pytest

import dvo_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vga_stream():
    from dvo_amd import synthetic
    return synthetic.make_stream(9, 640, 480, seed=5, device="cuda")


def test_batch_matches_ground_truth_and_oracle(vga_stream):
    from dvo_amd.batch import BatchAligner, compose_trajectory, absolute_trajectory_error
    from dvo_amd import synthetic
    st = vga_stream
    al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, max_iterations=10)
    res = al.align_host(st["gray"], st["depth"])
    al.close()
    assert (res["iters"] == 40).all()
    for k in range(8):
        gt = synthetic.relative_transform(st["T_wc"][k], st["T_wc"][k + 1])
        assert np.abs(res["T"][k] - gt[:3]).max() < 2e-3          # converges to the rendered motion
    # oracle on two of the pairs: same answer to float32-pose accuracy
    for k in (0, 5):
        est = O.gauss_newton(st["gray"][k], st["depth"][k], st["gray"][k + 1], st["K"], st["depth_scale"],
                             levels=4, max_iterations=10)
        # the oracle (like the reference) keeps float32 quaternions between iterations, the device float64
        np.testing.assert_allclose(res["xi"][k], est.log(), rtol=0, atol=3e-6)
    traj = compose_trajectory(res["T"], init_pose=st["T_wc"][0])
    assert absolute_trajectory_error(traj, st["T_wc"]) < 2e-3      # metres, 8 frames of ~1 cm steps


def test_device_ingest_equals_host_ingest(vga_stream):
    """dvo_upload_frames_device (batched, from HBM) builds bit-identical pyramids to per-frame host uploads."""
    import torch
    from dvo_amd.batch import BatchAligner
    st = vga_stream
    al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, max_iterations=10)
    g = torch.from_numpy(st["gray"]).cuda()
    d = torch.from_numpy(st["depth"].view(np.int16)).cuda()
    al.ingest_device(g.data_ptr(), d.data_ptr(), 9)
    al.ctx.synchronize()
    gp = O.build_pyramid(st["gray"][4], 4)
    dp = O.build_pyramid(st["depth"][4], 4)
    for lvl in range(4):
        got = al.ctx.download_level(4, lvl)
        assert np.array_equal(got["gray"], gp[lvl]) and np.array_equal(got["depth"], dp[lvl])
        gx, gy = O.sobel3(gp[lvl])
        assert np.array_equal(got["gx"], gx.astype(np.int16)) and np.array_equal(got["gy"], gy.astype(np.int16))
    al.solve_enqueue(9)
    res_dev = al.solve_fetch(9)
    res_host = al.align_host(st["gray"], st["depth"])
    al.close()
    assert np.array_equal(res_dev["xi"], res_host["xi"])            # same data, deterministic reduction


def test_back_to_back_asynchronous_uploads_on_one_context(vga_stream):
    """Several dvo_upload_frames_device calls enqueued without a sync in between (different slot lists, different
    frames): each call's slot-id list is read by the GPU later, from its own page-locked ring entry."""
    import torch
    import dvo_amd
    st = vga_stream
    g = torch.from_numpy(st["gray"]).cuda()
    d = torch.from_numpy(st["depth"].view(np.int16)).cuda()
    npx = 640 * 480
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=3, n_slots=9, max_pairs=1) as ctx:
        K = st["K"]
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
        order = [(np.array([7, 2], dtype=np.int32), 0), (np.array([0], dtype=np.int32), 2), (np.array([5, 1, 8], dtype=np.int32), 3),
                 (np.array([3, 4], dtype=np.int32), 6), (np.array([6], dtype=np.int32), 8), (np.array([7], dtype=np.int32), 4)]
        for slots, first in order:                                   # six uploads, nothing waits in between
            ctx.upload_frames_device(slots, g.data_ptr() + first * npx, d.data_ptr() + 2 * first * npx, 640, 480, 3)
        ctx.synchronize()
        expect = {2: 1, 0: 2, 5: 3, 1: 4, 8: 5, 3: 6, 4: 7, 6: 8, 7: 4}     # slot -> frame (slot 7 overwritten last)
        for slot, frame in expect.items():
            got = ctx.download_level(slot, 2, want=("gray", "depth"))
            assert np.array_equal(got["gray"], O.build_pyramid(st["gray"][frame], 3)[2]), (slot, frame)
            assert np.array_equal(got["depth"], O.build_pyramid(st["depth"][frame], 3)[2]), (slot, frame)


def test_huber_and_student_t_device_vs_oracle(vga_stream):
    from dvo_amd.batch import BatchAligner
    st = vga_stream
    for weighting, kw in (("huber", dict(weighting="huber")), ("student_t_reference", dict(use_weighter=True))):
        al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 3, max_pairs=8, max_iterations=6, weighting=weighting)
        res = al.align_host(st["gray"][:3], st["depth"][:3])
        al.close()
        for k in range(2):
            est = O.gauss_newton(st["gray"][k], st["depth"][k], st["gray"][k + 1], st["K"], st["depth_scale"],
                                 levels=3, max_iterations=6, **kw)
            np.testing.assert_allclose(res["xi"][k], est.log(), rtol=0, atol=2e-6)


def test_student_t_fast_path_is_bit_identical_to_the_plain_one(vga_stream, monkeypatch):
    """Student-t device solve with the reference's passes as written (DVO_ST_EXACT; the default one-pass moment form has its own
    test below): the second scale pass sums the residuals the first pass kept and every kernel
    derives its lambda from the per-workgroup sums of the passes before it; DVO_ST_SOLVE_PHASES keeps the cache but
    runs the control kernel after each pass; DVO_NO_RCACHE also recomputes the residuals (the original flow).  Same
    record->lane map and summation orders => identical poses."""
    from dvo_amd.batch import BatchAligner
    st = vga_stream
    out = []
    for env in (("DVO_ST_EXACT",), ("DVO_ST_SOLVE_PHASES",), ("DVO_NO_RCACHE", "DVO_ST_EXACT")):
        for k in ("DVO_ST_SOLVE_PHASES", "DVO_NO_RCACHE", "DVO_ST_EXACT"):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, "1")
        al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, max_iterations=6,
                          weighting="student_t_reference")
        out.append(al.align_host(st["gray"], st["depth"]))
        al.close()
    for o in out[1:]:
        assert np.array_equal(out[0]["xi"], o["xi"]) and np.array_equal(out[0]["err"], o["err"])


def test_student_t_moment_form_equals_the_three_pass_form(frames, vga_stream, monkeypatch):
    """The reference's Student-t at the per-iteration levels in ONE pass (the exact lambda from moments of the same pass, the weighted
    sums under a guessed lambda corrected to first order: include/dvo_hip.h) against the reference's passes as written
    (DVO_ST_EXACT): the same iteration counts and poses to 2e-8 on the reference's fixture pairs and on rendered pairs, and no
    pair had to be solved again."""
    from dvo_amd.batch import BatchAligner
    for gray, depth, K, scale in ((frames["gray"], frames["depth"], frames["K"], frames["depth_scale"]),
                                  (vga_stream["gray"], vga_stream["depth"], vga_stream["K"], vga_stream["depth_scale"])):
        out = []
        for exact in (False, True):
            monkeypatch.delenv("DVO_ST_EXACT", raising=False)
            if exact:
                monkeypatch.setenv("DVO_ST_EXACT", "1")
            al = BatchAligner(K, scale, 640, 480, 4, max_pairs=16, max_iterations=10, weighting="student_t_reference")
            out.append(al.align_host(gray[:10], depth[:10]))
            assert al.ctx.student_t_fallbacks() == 0 and al.ctx.student_t_fallback_pairs() == 0
            al.close()
        monkeypatch.delenv("DVO_ST_EXACT", raising=False)
        assert np.array_equal(out[0]["iters"], out[1]["iters"])
        assert np.max(np.abs(out[0]["xi"] - out[1]["xi"])) < 2e-8
        np.testing.assert_allclose(out[0]["err"], out[1]["err"], rtol=1e-6)


def test_student_t_result_does_not_depend_on_the_batch_size(frames):
    """The same frame pair solved in a batch of 16 and in a batch of 17 (the size at which the unweighted solve moves the 160x120
    level into the persistent kernel): with the reference's Student-t every level up to 160x120 runs the reference's IRLS loop in
    the persistent kernel whatever the batch size, so the poses agree to float32 summation noise."""
    from dvo_amd.batch import BatchAligner
    res = []
    for n in (16, 17):
        gray = np.stack([frames["gray"][k % 2] for k in range(n + 1)])
        depth = np.stack([frames["depth"][k % 2] for k in range(n + 1)])
        al = BatchAligner(frames["K"], frames["depth_scale"], 640, 480, 4, max_pairs=n, max_iterations=10, weighting="student_t_reference")
        res.append(al.align_host(gray, depth))
        al.close()
    assert np.array_equal(res[0]["iters"][:2], res[1]["iters"][:2])
    assert np.max(np.abs(res[0]["xi"][:2] - res[1]["xi"][:2])) < 1e-7


def _tripping_pair(seed=5, n_flip=40, w=640, h=480):
    """Two identical frames except `n_flip` inverted pixels: almost every residual is zero, the summed Student-t scale is small and
    lambda is not - the regime in which the one-pass forms of the reference's weighter do not apply."""
    from dvo_amd import synthetic
    st = synthetic.make_stream(2, w, h, seed=seed)
    gray = np.stack([st["gray"][0], st["gray"][0]]).copy()
    rng = np.random.default_rng(3)
    ys, xs = rng.integers(40, h - 40, n_flip), rng.integers(40, w - 40, n_flip)
    gray[1, ys, xs] = 255 - gray[1, ys, xs]
    depth = np.stack([st["depth"][0], st["depth"][0]])
    return st, gray, depth


def test_student_t_moment_form_falls_back_when_its_bound_fails():
    """Two identical frames except a handful of saturated pixels: the series the one-pass form relies on does not apply.  The pair is
    flagged, solved again with the passes as written (a compacted batch of one), and the result is bit-identical to a context that
    never used the one-pass form."""
    import os
    from dvo_amd.batch import BatchAligner
    st, gray, depth = _tripping_pair()
    res = []
    for exact in (False, True):
        os.environ.pop("DVO_ST_EXACT", None)
        if exact:
            os.environ["DVO_ST_EXACT"] = "1"
        try:
            al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 2, max_pairs=1, max_iterations=4, weighting="student_t_reference")
            res.append(al.align_host(gray, depth))
            assert al.ctx.student_t_fallbacks() == (0 if exact else 1)
            assert al.ctx.student_t_fallback_pairs() == (0 if exact else 1)
            al.close()
        finally:
            os.environ.pop("DVO_ST_EXACT", None)
    assert np.array_equal(res[0]["xi"], res[1]["xi"]) and np.array_equal(res[0]["iters"], res[1]["iters"])
    # (no comparison with the NumPy restatement here: with a mean weighted residual of 6e-4 at 320x240 the stop rule |d err| < 1e-6
    # decides on float32 noise; the loop itself is compared with the oracle in the two tests below)


def test_student_t_fallback_is_per_pair(vga_stream):
    """A batch of 17 pairs, one of which trips the one-pass bound: only that pair is solved again (as a compacted batch of one, in
    the passes as written: bit-identical to solving it alone in a DVO_ST_EXACT context); the 16 others keep the results of the
    one-pass solve bit for bit (what the same 16 pairs give in a batch without the tripping pair)."""
    import os
    import dvo_amd
    from dvo_amd import _cabi
    st = vga_stream
    _, tgray, tdepth = _tripping_pair()
    K = st["K"]
    prev = np.concatenate([np.arange(8), np.arange(1, 9)]).astype(np.int32)
    cur = np.concatenate([np.arange(1, 9), np.arange(8)]).astype(np.int32)
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=11, max_pairs=17) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
        for k in range(9):
            ctx.upload_frame(k, st["gray"][k], st["depth"][k], 4)
        ctx.upload_frame(9, tgray[0], tdepth[0], 4)
        ctx.upload_frame(10, tgray[1], tdepth[1], 4)
        clean = ctx.solve_batch(prev, cur, levels=4, max_iterations=10, weight_mode=_cabi.WEIGHT_STUDENT_T)
        assert ctx.student_t_fallback_pairs() == 0
        # the tripping pair in the MIDDLE of the batch
        p17 = np.concatenate([prev[:5], [9], prev[5:]]).astype(np.int32)
        c17 = np.concatenate([cur[:5], [10], cur[5:]]).astype(np.int32)
        mixed = ctx.solve_batch(p17, c17, levels=4, max_iterations=10, weight_mode=_cabi.WEIGHT_STUDENT_T)
        assert ctx.student_t_fallbacks() == 1 and ctx.student_t_fallback_pairs() == 1
    keep = np.r_[0:5, 6:17]
    for key in ("xi", "T", "iters", "err"):
        assert np.array_equal(mixed[key][keep], clean[key]), key
    os.environ["DVO_ST_EXACT"] = "1"
    try:
        with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=2, max_pairs=1) as ctx:
            ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
            ctx.upload_frame(0, tgray[0], tdepth[0], 4)
            ctx.upload_frame(1, tgray[1], tdepth[1], 4)
            alone = ctx.solve_batch([0], [1], levels=4, max_iterations=10, weight_mode=_cabi.WEIGHT_STUDENT_T)
    finally:
        os.environ.pop("DVO_ST_EXACT", None)
    for key in ("xi", "T", "iters", "err"):
        assert np.array_equal(mixed[key][5], alone[key][0]), key


def test_student_t_irls_of_another_shape_is_solved_by_the_reference_loop():
    """Identical frames except THREE inverted pixels: the reference's IRLS (weighter/t_weighter.py:24-32) needs a third pass at the
    finest level (lambda 0.04 -> ~2e-3 -> ~1e-4 -> ~1e-5), which neither the one-pass forms nor the two scale passes of the
    per-iteration levels evaluate.  Both detect it; the pair is solved again with every level in the persistent kernel, which runs
    the loop as written: same answer from a default context and a DVO_ST_EXACT one, and that of the NumPy restatement of the loop."""
    import os
    from dvo_amd import synthetic
    from dvo_amd.batch import BatchAligner
    st = synthetic.make_stream(2, 640, 480, seed=7)
    g0 = st["gray"][0].copy()
    depth = np.stack([st["depth"][0], st["depth"][0]])
    ys, xs = np.nonzero(depth[0][60:420, 60:580])
    pick = [(int(ys[i]) + 60, int(xs[i]) + 60) for i in (1000, 50000, 100000)]
    g1 = g0.copy()
    for (y, x) in pick:
        g0[y, x] = 0
        g1[y, x] = 255
    gray = np.stack([g0, g1])
    res, traces = [], []
    for exact in (False, True):
        os.environ.pop("DVO_ST_EXACT", None)
        if exact:
            os.environ["DVO_ST_EXACT"] = "1"
        try:
            al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 1, max_pairs=1, max_iterations=3, weighting="student_t_reference")
            al.ctx.trace_enable(3)
            res.append(al.align_host(gray, depth))
            traces.append(al.ctx.trace_read(1))
            assert al.ctx.student_t_fallback_pairs() == 1
            al.close()
        finally:
            os.environ.pop("DVO_ST_EXACT", None)
    assert np.array_equal(res[0]["xi"], res[1]["xi"]) and np.array_equal(res[0]["iters"], res[1]["iters"])
    # the oracle's loop on the first iteration really takes three passes
    lv = O.PairLevel(g0, depth[0], g1, O.level_intrinsics(st["K"], 0), st["depth_scale"])
    r, _, _, _ = O.residuals_and_jacobian(lv, np.eye(4, dtype=np.float32))
    lam, lams = 1.0 / 25.0, []
    for _ in range(6):
        cur = 1.0 / O.student_t_scale_sum((r * r).astype(np.float64), lam)
        lams.append(cur)
        if abs(cur - lam) < 1e-3:
            break
        lam = cur
    assert len(lams) == 3, lams
    # every iteration of the re-solve (its trace replaces the first solve's rows), teacher-forced at the pose the device used,
    # against the weighter loop as written (poses of ~1e-6 are not compared: the Lie classes drop rotations below 1e-6 and the
    # stop rule works on float32 noise here)
    tr = traces[0]
    n_it = int(res[0]["iters"][0])
    assert n_it >= 1
    for j in range(n_it):
        T = np.eye(4, dtype=np.float32)
        T[:3, :] = tr[0, j, 38:50].reshape(3, 4).astype(np.float32)
        rj, Jj, _, _ = O.residuals_and_jacobian(lv, T)
        w, _ = O.student_t_weights(rj)
        Ho, bo, _, no = O.normal_equations(rj, Jj, w)
        assert no == int(round(tr[0, j, 28]))
        inc = O.solve_increment(Ho, bo)
        # (3e-5, not the 1e-5 of ordinary pairs: the right-hand side comes from THREE pixels, so the float32 rounding of their
        # Jacobian rows - 2e-6 of the column scale, tests/test_hip_parity.py - does not average out over the image)
        assert np.linalg.norm(tr[0, j, 32:38] - inc) / np.linalg.norm(inc) < 3e-5, j


def test_student_t_static_camera_with_sensor_noise_needs_no_fallback():
    """What every TUM sequence starts with: the camera at rest, two exposures of one scene that differ by sensor noise (grey levels
    +-2, depth +-2 raw units).  Every residual is small but none is zero, the summed scale is large and the one-pass forms apply:
    no pair is solved again, and the pose equals the NumPy restatement of the reference's weighter loop."""
    from dvo_amd import synthetic
    from dvo_amd.batch import BatchAligner
    st = synthetic.make_stream(1, 640, 480, seed=9)
    rng = np.random.default_rng(11)
    g = st["gray"][0].astype(np.float32)
    d = st["depth"][0].astype(np.int32)
    n = 5
    gray = np.stack([np.clip(np.rint(g + rng.normal(0, 2.0, g.shape)), 0, 255).astype(np.uint8) for _ in range(n)])
    depth = np.stack([np.where(d > 0, np.clip(d + rng.integers(-2, 3, d.shape), 1, 65535), 0).astype(np.uint16) for _ in range(n)])
    al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=n - 1, max_iterations=10, weighting="student_t_reference")
    res = al.align_host(gray, depth)
    assert al.ctx.student_t_fallbacks() == 0 and al.ctx.student_t_fallback_pairs() == 0
    al.close()
    assert np.abs(res["xi"]).max() < 2e-3                                      # at rest
    for k in (0, n - 2):
        est = O.gauss_newton(gray[k], depth[k], gray[k + 1], st["K"], st["depth_scale"], levels=4, max_iterations=10, use_weighter=True)
        np.testing.assert_allclose(res["xi"][k], est.log(), rtol=0, atol=2e-6)


def test_student_t_fallback_refuses_frames_that_changed_since_enqueue():
    """Split enqueue / fetch: the per-pair fallback reads the frames at fetch time.  A slot uploaded again in between is detected
    (upload epochs) and the fetch fails with DVO_ERR_BAD_STATE instead of solving the flagged pair on other frames; without a
    flagged pair the same sequence of calls is fine."""
    import dvo_amd
    from dvo_amd import _cabi
    st, gray, depth = _tripping_pair()
    K = st["K"]
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=4, n_slots=3, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
        ctx.upload_frame(0, gray[0], depth[0], 4)
        ctx.upload_frame(1, gray[1], depth[1], 4)
        args = (4, 10, 1e-6, 0, _cabi.VARIANT_V1, _cabi.WEIGHT_STUDENT_T, 0.0, None)
        ctx.solve_batch_enqueue(np.array([0], np.int32), np.array([1], np.int32), *args)
        ctx.upload_frame(1, gray[0], depth[0], 4)                    # the frame the flagged pair needs is gone
        with pytest.raises(dvo_amd.DVOHipError) as e:
            ctx.solve_batch_fetch(1)
        assert e.value.code == -5 and "uploaded again" in str(e.value)
        # ... and a second fetch of a solve whose first fetch solved pairs again is refused (the device buffers hold the re-solve)
        ctx.upload_frame(1, gray[1], depth[1], 4)
        ctx.solve_batch_enqueue(np.array([0], np.int32), np.array([1], np.int32), *args)
        ctx.solve_batch_fetch(1)
        assert ctx.student_t_fallback_pairs() >= 1
        with pytest.raises(dvo_amd.DVOHipError) as e:
            ctx.solve_batch_fetch(1)
        assert e.value.code == -5 and "fetch once per enqueue" in str(e.value)
        # an ordinary pair: nothing is flagged, the same call order is legal
        from dvo_amd import synthetic
        st2 = synthetic.make_stream(2, 640, 480, seed=5)
        ctx.upload_frame(0, st2["gray"][0], st2["depth"][0], 4)
        ctx.upload_frame(1, st2["gray"][1], st2["depth"][1], 4)
        ctx.solve_batch_enqueue(np.array([0], np.int32), np.array([1], np.int32), *args)
        ctx.upload_frame(2, gray[0], depth[0], 4)
        ctx.upload_frame(1, gray[0], depth[0], 4)
        res = ctx.solve_batch_fetch(1)
        assert np.isfinite(res["xi"]).all()


def test_config3_1280x960_five_levels_huber():
    """BASELINE config 3 shape: 1280x960, 5-level pyramid, Huber weighting — converges to the rendered motion."""
    from dvo_amd import synthetic
    from dvo_amd.batch import BatchAligner
    st = synthetic.make_stream(3, 1280, 960, seed=11, device="cuda")
    al = BatchAligner(st["K"], st["depth_scale"], 1280, 960, 5, max_pairs=2, max_iterations=10, weighting="huber")
    res = al.align_host(st["gray"], st["depth"])
    al.close()
    for k in range(2):
        gt = synthetic.relative_transform(st["T_wc"][k], st["T_wc"][k + 1])
        assert np.abs(res["T"][k] - gt[:3]).max() < 2e-3
        assert res["iters"][k] > 25


def test_batched_host_upload_and_streamed_aligner(vga_stream):
    """dvo_upload_frames (two large H2D copies, page-locked memory) + StreamedAligner.align_host == per-pair results."""
    from dvo_amd import _cabi
    from dvo_amd.batch import BatchAligner, StreamedAligner
    st = vga_stream
    g = _cabi.pinned_array(st["gray"].shape, np.uint8)
    d = _cabi.pinned_array(st["depth"].shape, np.uint16)
    g[...] = st["gray"]
    d[...] = st["depth"]
    ref = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, max_iterations=10)
    r0 = ref.align_host(st["gray"], st["depth"])
    lvl2 = ref.ctx.download_level(3, 2)
    r0_rev = ref.align_host(st["gray"][::-1], st["depth"][::-1])
    sa = StreamedAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, n_streams=2, max_iterations=10)
    r1 = sa.align_host(g, d)
    got = sa.parts[0].ctx.download_level(3, 2)
    # back-to-back uploads through the shared copy stream reuse each slice's staging area: the second batch (frames in
    # reverse order) must not overtake the ingest kernels of the first
    g2 = _cabi.pinned_array(st["gray"].shape, np.uint8)
    d2 = _cabi.pinned_array(st["depth"].shape, np.uint16)
    g2[...] = st["gray"][::-1]
    d2[...] = st["depth"][::-1]
    sa.align_host(g, d)
    r2 = sa.align_host(g2, d2)
    sa.close()
    ref.close()
    np.testing.assert_allclose(r2["xi"], r0_rev["xi"], rtol=0, atol=1e-9)
    for k in ("gray", "depth", "gx", "gy"):
        assert np.array_equal(got[k], lvl2[k])
    assert r1["xi"].shape == (8, 6)
    # slices of 4 pairs per stream vs one batch of 8: same data, at most last-bit differences (lane grouping)
    np.testing.assert_allclose(r1["xi"], r0["xi"], rtol=0, atol=1e-9)


def test_pipelined_windows_equal_window_by_window_alignment(vga_stream):
    """StreamedAligner.submit_device / drain (a slice's next window enqueued behind the fetch of its previous result, no drain of the GPU
    between windows) returns, one call later, exactly what align_device returns for the same window - also when consecutive windows hold
    different frames - and refuses a change of window size with a window in flight."""
    import torch
    from dvo_amd.batch import StreamedAligner
    n = 9
    g0 = torch.as_tensor(vga_stream["gray"][:n]).cuda().contiguous()
    d0 = torch.as_tensor(vga_stream["depth"][:n].view(np.int16)).cuda().contiguous()
    g1, d1 = torch.flip(g0, dims=[0]).contiguous(), torch.flip(d0, dims=[0]).contiguous()      # the same frames played backwards
    al = StreamedAligner(vga_stream["K"], vga_stream["depth_scale"], 640, 480, 4, max_pairs=n - 1, n_streams=3, max_iterations=4)
    try:
        want = [al.align_device(g.data_ptr(), d.data_ptr(), n) for g, d in ((g0, d0), (g1, d1), (g0, d0))]
        assert al.drain() is None
        got = [al.submit_device(g.data_ptr(), d.data_ptr(), n) for g, d in ((g0, d0), (g1, d1), (g0, d0))]
        with pytest.raises(ValueError):
            al.submit_device(g0.data_ptr(), d0.data_ptr(), n - 2)
        got.append(al.drain())
        assert got[0] is None and al.drain() is None
        for w, g in zip(want, got[1:]):
            for k in ("T", "xi", "iters", "err"):
                assert np.array_equal(w[k], g[k]), k
        assert not np.array_equal(want[0]["xi"], want[1]["xi"])
    finally:
        al.close()


@pytest.mark.parametrize("colour", [False, True])
def test_pipelined_host_windows_equal_window_by_window_alignment(vga_stream, colour):
    """sequence.align_windows over a StreamedAligner (submit_host / drain: windows of page-locked host frames pipelined through the
    slices, a shorter last window preceded by a drain inside submit_host) returns, in sequence order, exactly what aligning every
    window by itself returns; frames copied into the ring by several threads."""
    from dvo_amd import _cabi, sequence
    from dvo_amd.batch import StreamedAligner
    st = vga_stream
    n, win = 9, 3
    assert st["gray"].shape[0] >= n
    gray, depth = st["gray"][:n], st["depth"][:n]
    img = np.repeat(gray[..., None], 3, axis=3) if colour else gray
    al = StreamedAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=win, n_streams=2, max_iterations=3)
    try:
        want = {"T": [], "iters": [], "err": []}
        for lo in range(0, n - 1, win):                                 # windows of 3 pairs: frames [0..3], [3..6], [6..8]
            hi = min(lo + win, n - 1)
            r = al.align_host_bgr(np.ascontiguousarray(img[lo:hi + 1]), depth[lo:hi + 1]) if colour else al.align_host(gray[lo:hi + 1], depth[lo:hi + 1])
            for k in want:
                want[k].append(r[k])
        fw = sequence.FrameWindows(((img[k], depth[k]) for k in range(n)), win, alloc=_cabi.pinned_array, copy_threads=3)
        got = sequence.align_windows(al, fw, colour)
        assert got["T"].shape[0] == n - 1
        for k in want:
            assert np.array_equal(np.concatenate(want[k]), got[k]), k
    finally:
        al.close()


def test_rccl_pose_gather_through_the_c_abi():
    """dvo_comm_* (RCCL via dlopen): world of one rank on the test box — unique id, communicator, all-gather."""
    from dvo_amd import _cabi
    uid = _cabi.RcclPoseGather.unique_id()
    assert len(uid) == 128
    comm = _cabi.RcclPoseGather(device=0, rank=0, nranks=1, unique_id=uid)
    rec = np.arange(5 * 16, dtype=np.float32).reshape(5, 16)
    out = comm.allgather(rec)
    assert np.array_equal(out, rec)
    out = comm.allgather(rec[:2] * 3)
    assert np.array_equal(out, rec[:2] * 3)
    comm.close()


def test_contexts_release_their_device_memory(vga_stream):
    """Create / use / destroy cycles (incl. the lazily allocated Student-t cache, batch staging, graphs, peer copy stream)
    leave the free device memory where it was."""
    import torch
    from dvo_amd.batch import StreamedAligner
    st = vga_stream
    torch.cuda.synchronize()
    free0 = None
    for cycle in range(6):
        sa = StreamedAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, n_streams=2, max_iterations=3,
                             weighting="student_t_reference" if cycle % 2 else "none")
        sa.align_host(st["gray"], st["depth"])
        sa.close()
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info()[0]
        if cycle == 1:
            free0 = free                              # after the first cycles: runtime pools are warm
        if cycle > 1:
            assert free >= free0 - (32 << 20), (cycle, free0 - free)


@pytest.mark.parametrize("max_increased", [0, 2])
def test_error_increase_budget_on_the_device_vs_oracle(max_increased):
    """max_increased_steps_allowed (base_robust_dvo.py:222-224, no rollback: quirk Q8) on the device-side control: a
    pair with a large motion (fixture frames 0 -> 4) and Huber weights, whose weighted error does go up on the way;
    iteration count and final twist against the oracle."""
    from pathlib import Path
    from dvo_amd.batch import BatchAligner
    f = np.load(Path(__file__).parent / "golden" / "frames.npz")
    g = f["gray"][[0, 4]][:, ::2, ::2].copy()                     # 320x240: the oracle stays quick
    d = f["depth"][[0, 4]][:, ::2, ::2].copy()
    K = O.level_intrinsics(f["intrinsics"], 1)
    ds = float(f["depth_scale"])
    rec = []
    est = O.gauss_newton(g[0], d[0], g[1], K, ds, levels=3, max_iterations=12, weighting="huber",
                         max_increased_steps_allowed=max_increased, record=rec)
    al = BatchAligner(K, ds, 320, 240, 3, max_pairs=1, max_iterations=12, weighting="huber",
                      max_increased_steps_allowed=max_increased)
    res = al.align_host(g, d)
    al.close()
    assert int(res["iters"][0]) == len(rec)
    np.testing.assert_allclose(res["xi"][0], est.log(), rtol=0, atol=5e-6)


# ------------------------------------------------------------------------------------------------
# Student-t: the mean-based ("proper") form and the IRLS loop on small / low-residual inputs
# ------------------------------------------------------------------------------------------------
def test_student_t_mean_weights_vs_oracle(vga_stream):
    """DVO_WEIGHT_STUDENT_T_MEAN through the C ABI: the scale statistics (sum AND count) of one pass and the normal
    equations under the converged lambda, against oracle.student_t_mean_weights (build-defined: parity unpinned)."""
    import dvo_amd
    from dvo_amd import _cabi
    st = vga_stream
    K = np.asarray(st["K"], dtype=np.float32)
    T = O.Pose.from_xi(np.array([0.002, -0.001, 0.003, 0.001, -0.002, 0.0015])).matrix()
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=3, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], st["depth_scale"])
        ctx.upload_frames([0, 1], st["gray"][:2], st["depth"][:2], 3)
        for lvl in (2, 0):
            lv = O.PairLevel(O.build_pyramid(st["gray"][0], 3)[lvl], O.build_pyramid(st["depth"][0], 3)[lvl],
                             O.build_pyramid(st["gray"][1], 3)[lvl], O.level_intrinsics(K, lvl), st["depth_scale"])
            r, J, _, _ = O.residuals_and_jacobian(lv, T)
            r2 = (r.astype(np.float32) ** 2).astype(np.float64).reshape(-1)
            s, n = ctx.weighted_r2_stats(lvl, [0], [1], T[None], [0.04])[0]
            assert n == r.shape[0]
            assert abs(s - O.student_t_scale_sum(r2, 0.04)) <= 2e-6 * s
            w, lam = O.student_t_mean_weights(r, 0.04)
            assert 1e-5 < lam < 1.0 and w.min() < 0.9 < w.max()          # a real robust weighting, unlike the quirk form (w == 1.2)
            rec = ctx.normal_equations(lvl, [0], [1], T[None], _cabi.VARIANT_V1, _cabi.WEIGHT_STUDENT_T_MEAN, [lam])[0]
            H, b, swr2, nn, _ = _cabi.unpack_record(rec)
            Ho, bo, so, no = O.normal_equations(r, J, w)
            assert nn == no
            assert np.linalg.norm(H - Ho) < 2e-6 * np.linalg.norm(Ho) and abs(swr2 - so) < 2e-6 * so
            inc, inco = np.linalg.solve(H, b), O.solve_increment(Ho, bo)
            assert np.linalg.norm(inc - inco) < 1e-5 * np.linalg.norm(inco)


@pytest.mark.parametrize("solver", ["host", "device"])
def test_student_t_mean_full_solve_vs_oracle(vga_stream, solver):
    """weighting="student_t" (mean-based scale, warm-started IRLS of at most 4 passes) over 3 levels: host-solve and
    device-solve paths (persistent kernel on the coarse level, scale passes + control launches on the finer ones) against
    the oracle's Gauss-Newton loop with the same definition."""
    import dvo_amd
    st = vga_stream
    cam = dvo_amd.RGBDCameraModel(st["K"], st["depth_scale"])
    dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=3, max_iterations=6, solver=solver,
                          weighting="student_t")
    dvo.step_gray(st["gray"][0], st["depth"][0])
    tf = dvo.step_gray(st["gray"][1], st["depth"][1])
    est = O.gauss_newton(st["gray"][0], st["depth"][0], st["gray"][1], st["K"], st["depth_scale"], levels=3,
                         max_iterations=6, weighting="student_t")
    np.testing.assert_allclose(tf.log().reshape(6), est.log(), rtol=0, atol=3e-6)
    assert dvo.last_iterations == 18


def test_student_t_reference_irls_needs_more_than_two_passes_on_a_tiny_level():
    """The reference's IRLS loop (weighter/t_weighter.py:24-32) stops after two passes whenever the summed scale is large
    (any real image).  With five valid pixels and residuals of ~20 grey levels lambda is still moving by more than 1e-3
    after the second pass, so the loop takes three.  Such levels run in the persistent kernel, which loops like the
    reference instead of taking two fixed passes: the weighted error of the first iteration (it depends on the final
    lambda directly) and the poses must match the oracle, on the device-solve and the host-solve path."""
    import dvo_amd
    from dvo_amd import _cabi
    h, w = 8, 10
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    g1 = (120 + 8 * xx - 5 * yy).astype(np.uint8)
    g2 = g1.copy()
    depth = np.zeros((h, w), dtype=np.uint16)
    for k, (y, x) in enumerate(((3, 3), (3, 6), (4, 4), (5, 3), (4, 6))):
        depth[y, x] = 5000 + 100 * k
        g2[y - 1:y + 2, x - 1:x + 2] = np.clip(g1[y - 1:y + 2, x - 1:x + 2].astype(int) + (20 + 3 * k if k % 2 else -21 + 2 * k), 0, 255)
    K = np.array([[8.0, 0, 4.5], [0, 8.0, 3.5], [0, 0, 1]], dtype=np.float32)
    scale = 0.0002
    lv = O.PairLevel(g1, depth, g2, O.level_intrinsics(K, 0), scale)
    r, _, _, _ = O.residuals_and_jacobian(lv, np.eye(4, dtype=np.float32))
    r2 = (r.astype(np.float32) ** 2).astype(np.float64).reshape(-1)
    lam, passes = 1.0 / 25.0, 0
    for passes in range(1, 51):                           # the passes the reference's loop takes on this input
        cur = 1.0 / O.student_t_scale_sum(r2, lam)
        if abs(cur - lam) < 1e-3:
            break
        lam = cur
    assert passes == 3
    rec = []
    # one iteration: five points leave the 6x6 system rank-deficient, and the minimum-norm steps it then takes (~1 rad) make
    # any later iterate hypersensitive to rounding; the first iteration's weighted error and step are what pins lambda
    est = O.gauss_newton(g1, depth, g2, K, scale, levels=1, max_iterations=1, use_weighter=True, record=rec)
    with dvo_amd.HipContext(max_w=w, max_h=h, max_levels=1, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0.0, scale)
        ctx.upload_frame(0, g1, depth, 1)
        ctx.upload_frame(1, g2, depth, 1)
        ctx.trace_enable(4)
        res = ctx.solve_batch([0], [1], levels=1, max_iterations=1, weight_mode=_cabi.WEIGHT_STUDENT_T)
        tr = ctx.trace_read(1)[0]
    assert int(res["iters"][0]) == len(rec)
    for k, it in enumerate(rec):
        assert tr[k, 28] == it.n == 5
        assert abs(tr[k, 27] - it.sum_wr2) <= 1e-5 * it.sum_wr2, (k, tr[k, 27], it.sum_wr2)     # wrong lambda: off by ~3e-3
    # (the step itself is not compared: with five points the system is rank-deficient, and what a least-squares solver returns
    # along its near-null direction is decided by 1e-7 rounding differences in H)
    cam = dvo_amd.RGBDCameraModel(K, scale)
    dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=1, max_iterations=1, solver="host",
                          use_weighter=True, height=h, width=w)
    dvo.step_gray(g1, depth)
    dvo._build_pyramids(g2, depth)
    _, _, err, n = dvo.compute_normal_equations(dvo_amd.Se3.identity(), level=0)       # host path: the reference's loop in Python
    assert n == 5 and abs(err - rec[0].err) <= 1e-5 * rec[0].err
    assert est is not None


def test_batched_colour_ingest_bit_exact(frames):
    """dvo_upload_frames_bgr: n colour frames in one batched upload; BGR->gray and the depth clamp of
    core/base_dense_visual_odometry.py:58-59 on the device == oracle (bgr2gray + float64 clamp) == the per-frame
    dvo_upload_frame_bgr path, bit for bit, pyramids and valid-point lists included."""
    import dvo_amd
    rng = np.random.default_rng(4)
    bgr0 = frames["color_bgr_1"]
    n = 3
    bgr = np.stack([bgr0, np.roll(bgr0, 7, axis=1), rng.integers(0, 256, size=bgr0.shape, dtype=np.uint8)])
    depth = np.stack([frames["depth"][k].copy() for k in range(n)])
    scale, maxd = frames["depth_scale"], 5.0
    # every depth value around the clamp threshold appears (the device compares integers; the reference a float64 product)
    thr = int(maxd / scale)
    depth[0, 10, :41] = np.arange(thr - 20, thr + 21, dtype=np.uint16)
    depth[1, 100:120, 200:260] = 40000
    K = frames["K"]
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=3, n_slots=4, max_pairs=2) as ctx:
        ctx.set_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1], scale)
        ctx.upload_frames_bgr([0, 1, 2], bgr, depth, 3, maxd)
        ctx.upload_frame_bgr(3, bgr[1], depth[1], 3, maxd)
        ctx.synchronize()
        for k in range(n):
            gray = O.bgr2gray(bgr[k])
            dk = depth[k].copy()
            dk[(dk * scale) > maxd] = 0                                   # the reference's expression (float64 product)
            gp, dp = O.build_pyramid(gray, 3), O.build_pyramid(dk, 3)
            for lvl in range(3):
                got = ctx.download_level(k, lvl)
                assert np.array_equal(got["gray"], gp[lvl]) and np.array_equal(got["depth"], dp[lvl]), (k, lvl)
                gx, gy = O.sobel3(gp[lvl])
                assert np.array_equal(got["gx"], gx.astype(np.int16)) and np.array_equal(got["gy"], gy.astype(np.int16))
                pts = ctx.download_points(k, lvl)
                ys, xs = np.nonzero(dp[lvl])
                assert np.array_equal(pts["x"], xs) and np.array_equal(pts["y"], ys)
        a, b = ctx.download_level(1, 0), ctx.download_level(3, 0)
        assert np.array_equal(a["gray"], b["gray"]) and np.array_equal(a["depth"], b["depth"])
        assert (depth[0, 10, :41] != 0).all() and (ctx.download_level(0, 0, want=("depth",))["depth"][10, :41] == 0).sum() in (19, 20, 21)


def test_streamed_colour_alignment_equals_gray_alignment(vga_stream):
    """StreamedAligner.align_host_bgr (colour frames, page-locked, conversion on the device) gives the poses of the gray
    path on the gray-equivalent input (B = G = R), bit for bit."""
    from dvo_amd.batch import StreamedAligner
    from dvo_amd import _cabi
    st = vga_stream
    n = 9
    bgr = _cabi.pinned_array((n, 480, 640, 3), np.uint8)
    dep = _cabi.pinned_array((n, 480, 640), np.uint16)
    bgr[...] = st["gray"][:n, :, :, None]
    dep[...] = st["depth"][:n]
    al = StreamedAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, n_streams=2, max_iterations=6)
    a = al.align_host_bgr(bgr, dep, max_distance=1e9)
    b = al.align_host(st["gray"][:n], st["depth"][:n])
    al.close()
    assert np.array_equal(a["xi"], b["xi"]) and np.array_equal(a["iters"], b["iters"])


@pytest.mark.parametrize("colour", [False, True])
def test_run_sequence_fixture_end_to_end(tmp_path, golden_dir, colour):
    """tools/run_sequence.py --fixture: decoder thread -> page-locked windows -> batched (gray or colour) upload -> device
    solve -> report.json in the reference's shape + TUM trajectory; ATE equal to the reference run's on the same excerpt."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    from dvo_amd import tum
    from dvo_amd.batch import absolute_trajectory_error
    root = Path(__file__).resolve().parents[1]
    cmd = [sys.executable, str(root / "tools" / "run_sequence.py"), "--fixture", "--out", str(tmp_path), "--window", "4"]
    out = subprocess.run(cmd + (["--colour"] if colour else []), capture_output=True, text=True, timeout=600, cwd=str(root))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    g = np.load(golden_dir / "sequence.npz")
    f = np.load(golden_dir / "frames.npz")
    ref_ate = absolute_trajectory_error(g["pose_T"].astype(np.float64), f["gt_T"])
    assert abs(line["ate_rmse_m"] - ref_ate) < 1e-6 and abs(ref_ate - 4.1317e-3) < 1e-6
    assert line["frames"] == 10 and line["input"] == ("bgr" if colour else "gray") and line["type"] == "TEST"
    rep = json.loads((tmp_path / "report.json").read_text())
    np.testing.assert_allclose(np.array(rep["estimated_transforms"]), g["tf_xi"].reshape(10, 6), atol=3e-6)
    np.testing.assert_allclose(np.array(rep["estimated_poses"]), g["pose_xi"].reshape(10, 6), atol=3e-5)
    stamps, T = tum.read_tum_trajectory(tmp_path / "report.txt")
    np.testing.assert_allclose(T, np.array(rep["estimated_poses_matrix"]).reshape(10, 4, 4), atol=1e-5)
    assert len(stamps) == 10


def test_run_sequence_reference_cli_compat(tmp_path, golden_dir):
    """--reference-cli-compat: the colour fixture frames reach the BGR2GRAY ingest in RGB order, as in the reference's CLI
    (src/test_dvo.py:183 + core/base_dense_visual_odometry.py:58).  The gray plane the device builds is bit-identical to the one
    the reference run saw, and the estimated transform equals that run's (cli_compat.npz, oracle/refrun/make_golden.py)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    import dvo_amd
    c = np.load(golden_dir / "cli_compat.npz")
    f = np.load(golden_dir / "frames.npz")
    with dvo_amd.HipContext(max_w=640, max_h=480, max_levels=1, n_slots=2, max_pairs=1) as ctx:
        ctx.set_camera(500, 500, 320, 240, 0, float(f["depth_scale"]))
        ctx.upload_frame_bgr(0, c["rgb_1"], c["depth_1"].copy(), 1, 5.0)
        assert np.array_equal(ctx.download_level(0, 0, want=("gray",))["gray"], c["gray_cli_1"])
    root = Path(__file__).resolve().parents[1]
    cmd = [sys.executable, str(root / "tools" / "run_sequence.py"), "--fixture", "--reference-cli-compat", "--out", str(tmp_path), "--window", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(root))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["frames"] == 2 and line["reference_cli_compat"] is True and line["input"].startswith("rgb fed as bgr")
    rep = json.loads((tmp_path / "report.json").read_text())
    np.testing.assert_allclose(np.array(rep["estimated_transforms"])[1], c["tf_xi"].reshape(6), atol=3e-6)
    # the default (true BGR order) run of the same two frames gives a measurably different transform
    g = np.load(golden_dir / "pairs_l4.npz")
    assert np.max(np.abs(c["tf_xi"].reshape(6) - g["final_xi"][0].reshape(6))) > 1e-6


def test_variant_gn_batched_with_weights_vs_oracle(vga_stream):
    """The build's true Gauss-Newton mode through the batched device solve, unweighted and with the three weightings, against the
    oracle's loop of the same definition (parity unpinned: no reference counterpart), 3 levels x 3 iterations."""
    from dvo_amd.batch import BatchAligner
    st = vga_stream
    for weighting, kw in (("none", dict()), ("huber", dict(weighting="huber")), ("student_t", dict(weighting="student_t")),
                          ("student_t_reference", dict(use_weighter=True))):
        al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 3, max_pairs=8, max_iterations=3, weighting=weighting,
                          reference_variant="gn")
        res = al.align_host(st["gray"][:3], st["depth"][:3])
        al.close()
        for k in range(2):
            est = O.gauss_newton(st["gray"][k], st["depth"][k], st["gray"][k + 1], st["K"], st["depth_scale"],
                                 levels=3, max_iterations=3, variant="gn", **kw)
            np.testing.assert_allclose(res["xi"][k], est.log(), rtol=0, atol=5e-6, err_msg=weighting)


def test_level_iteration_schedule_vs_oracle(vga_stream):
    """dvo_set_level_iterations: per-level iteration counts (finest first, 0 skips a level) on the batched device solve and on
    the host loop of the drop-in class, against the oracle's loop with the same schedule; clearing it restores max_iterations."""
    import dvo_amd
    from dvo_amd.batch import BatchAligner
    st = vga_stream
    for variant, kw in (("gn", dict(reference_variant="gn")), ("v1", dict())):
        for sched in ([1, 2, 3, 2], [0, 2, 1, 3]):
            al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, max_iterations=7, level_iterations=sched, **kw)
            res = al.align_host(st["gray"][:3], st["depth"][:3])
            for k in range(2):
                est = O.gauss_newton(st["gray"][k], st["depth"][k], st["gray"][k + 1], st["K"], st["depth_scale"], levels=4,
                                     max_iterations=7, variant=variant, level_iterations=sched)
                np.testing.assert_allclose(res["xi"][k], est.log(), rtol=0, atol=5e-6, err_msg=str((variant, sched)))
                assert res["iters"][k] <= sum(sched)
            al.ctx.set_level_iterations(None)                                     # back to max_iterations = 7 at every level
            res7 = al.align_host(st["gray"][:3], st["depth"][:3])
            al.close()
            est7 = O.gauss_newton(st["gray"][0], st["depth"][0], st["gray"][1], st["K"], st["depth_scale"], levels=4,
                                  max_iterations=7, variant=variant)
            np.testing.assert_allclose(res7["xi"][0], est7.log(), rtol=0, atol=5e-6)
    cam = dvo_amd.RGBDCameraModel(st["K"], st["depth_scale"])
    est = O.gauss_newton(st["gray"][0], st["depth"][0], st["gray"][1], st["K"], st["depth_scale"], levels=4, max_iterations=9,
                         variant="gn", level_iterations=[1, 2, 2, 3])
    for solver in ("host", "device"):
        dvo = dvo_amd.get_dvo("robust-dvo", cam, dvo_amd.Se3.identity(), levels=4, max_iterations=9, solver=solver,
                              reference_variant="gn", level_iterations=[1, 2, 2, 3])
        dvo.step_gray(st["gray"][0], st["depth"][0])
        tf = dvo.step_gray(st["gray"][1], st["depth"][1])
        np.testing.assert_allclose(tf.log().reshape(6), est.log(), rtol=0, atol=5e-6)


def test_zero_iterations_return_the_initial_pose(vga_stream):
    """max_iterations = 0 (and an all-zero schedule): the reference's level loop does not run (base_robust_dvo.py:161), the
    estimate is the initial guess and no iteration is counted — for small and large batches (persistent kernel or not)."""
    from dvo_amd.batch import BatchAligner
    st = vga_stream
    for n_pairs in (2, 8):
        for kw in (dict(max_iterations=0), dict(max_iterations=5, level_iterations=[0, 0, 0, 0])):
            al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, **kw)
            res = al.align_host(st["gray"][:n_pairs + 1], st["depth"][:n_pairs + 1])
            al.close()
            assert np.all(res["iters"] == 0)
            np.testing.assert_allclose(res["xi"], 0.0, atol=1e-12)
            np.testing.assert_allclose(res["T"], np.tile(np.eye(4, dtype=np.float32)[:3], (n_pairs, 1, 1)), atol=1e-7)


def test_persistent_small_level_kernel_vs_per_iteration_launches(vga_stream, monkeypatch):
    """The coarse levels (<= 160x120) run in one persistent launch per batch (k_gn_small), a workgroup per pair; with
    DVO_SMALL_MAX_PIX=0 every level takes the per-iteration launches (k_reduce + k_solve).  Same per-record arithmetic,
    different summation grouping (the float32 per-lane pre-sums cover different records): poses agree to a few 1e-8, for every workgroup width, weighting and
    variant, and iteration counts are equal."""
    from dvo_amd.batch import BatchAligner
    st = vga_stream
    for kw in (dict(), dict(weighting="huber"), dict(weighting="student_t_reference"), dict(approximate_image2_gradient=True),
               dict(reference_variant="gn"), dict(reference_variant="gn", weighting="student_t")):
        out = []
        for env in ({"DVO_SMALL_MAX_PIX": "0"}, {"DVO_SMALL_THREADS": "256"}, {"DVO_SMALL_THREADS": "512"},
                    {"DVO_SMALL_THREADS": "1024"}, {}):
            for k in ("DVO_SMALL_MAX_PIX", "DVO_SMALL_THREADS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            al = BatchAligner(st["K"], st["depth_scale"], 640, 480, 4, max_pairs=8, max_iterations=6, **kw)
            out.append(al.align_host(st["gray"], st["depth"]))
            al.close()
        for o in out[1:]:
            if kw.get("reference_variant") == "gn":
                # a converging loop ends on |d err| < tolerance: summation-order noise of 1e-8 moves that decision by an iteration
                # or two, each of them a step of the size the loop was about to stop at
                assert np.abs(o["iters"] - out[0]["iters"]).max() <= 3
                np.testing.assert_allclose(o["xi"], out[0]["xi"], rtol=0, atol=5e-6)
                continue
            assert np.array_equal(o["iters"], out[0]["iters"])
            np.testing.assert_allclose(o["xi"], out[0]["xi"], rtol=0, atol=2e-7)      # measured: up to 2.8e-8


def test_run_sequence_on_a_tum_layout_dataset(tmp_path):
    """tools/run_sequence.py --tum-dir on a TUM-RGB-D-layout dataset written to disk (rgb/*.png, depth/*.png 16-bit, rgb.txt /
    depth.txt / groundtruth.txt with unsynchronised stamps, camera yaml): association as src/test_dvo.py:113-170, PNG
    decoding in the decoder pool, batched colour ingest, device solve, report + TUM trajectory.  The frames are a rendered
    stream with known poses, so the trajectory error checks the whole chain; also through the rank launcher (--spawn)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    from PIL import Image
    from dvo_amd import synthetic, tum
    root = Path(__file__).resolve().parents[1]
    n = 14
    st = synthetic.make_stream(n, 640, 480, seed=21, device="cuda")
    data = tmp_path / "rgbd_dataset_synthetic"
    (data / "rgb").mkdir(parents=True)
    (data / "depth").mkdir()
    t0 = 1305031102.0
    with (data / "rgb.txt").open("w") as fr, (data / "depth.txt").open("w") as fd, (data / "groundtruth.txt").open("w") as fg:
        fr.write("# color images\n# file: 'synthetic'\n# timestamp filename\n")
        fd.write("# depth maps\n# timestamp filename\n")
        fg.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
        for k in range(n):
            tr, td = t0 + k / 30.0, t0 + k / 30.0 + 0.004            # depth lags the colour image by 4 ms, as on a Kinect
            Image.fromarray(np.repeat(st["gray"][k][..., None], 3, axis=2)).save(data / "rgb" / ("%.6f.png" % tr))
            Image.fromarray(st["depth"][k].astype(np.uint16)).save(data / "depth" / ("%.6f.png" % td))
            fr.write("%.6f rgb/%.6f.png\n" % (tr, tr))
            fd.write("%.6f depth/%.6f.png\n" % (td, td))
            q = tum.matrix_to_quat_xyzw(st["T_wc"][k][:3, :3])
            fg.write("%.6f %r %r %r %r %r %r %r\n" % ((tr + 0.002,) + tuple(float(v) for v in st["T_wc"][k][:3, 3]) + tuple(float(v) for v in q)))
    cam = tmp_path / "camera.yaml"
    K = np.asarray(st["K"], dtype=np.float64)
    cam.write_text("intrinsics:\n" + "".join("  - [%r, %r, %r]\n" % tuple(float(v) for v in row) for row in K) +
                   "depth_scale: %r\n" % float(st["depth_scale"]))
    for extra in ([], ["--gpus", "1", "--spawn"], ["--decoder-processes", "3", "--window", "4"]):      # threads / launcher / worker processes
        out_dir = tmp_path / ("out%d" % len(extra))
        cmd = [sys.executable, str(root / "tools" / "run_sequence.py"), "--tum-dir", str(data), "--camera", str(cam),
               "--out", str(out_dir), "--window", "5", "--decoders", "4"] + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root))
        assert out.returncode == 0, out.stderr[-3000:]
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert line["type"] == "TUM" and line["frames"] == n and line["input"] == "bgr"
        assert line["ate_rmse_m"] < 2e-3 and line["rpe_trans_rmse_m"] < 5e-4        # ~1 cm steps recovered to a few 0.01 mm each
        rep = json.loads((out_dir / "report.json").read_text())
        assert len(rep["estimated_poses"]) == n and abs(rep["timestamps"][3] - (t0 + 3 / 30.0 + 0.002)) < 1e-5
        stamps, T = tum.read_tum_trajectory(out_dir / "report.txt")
        assert len(stamps) == n and np.abs(T[0] - st["T_wc"][0]).max() < 1e-5
