"""Sequence front end (dvo_amd/sequence.py), CPU part: windows of frames decoded by a background thread, and the report
files in the reference's shape (src/test_dvo.py:323-345)."""
import json

import numpy as np
import pytest

from dvo_amd import sequence, tum


def _frames(n, h=6, w=8, colour=False):
    for k in range(n):
        img = np.full((h, w, 3) if colour else (h, w), k, dtype=np.uint8)
        yield img, np.full((h, w), 1000 + k, dtype=np.uint16)


@pytest.mark.parametrize("n,window", [(10, 4), (9, 4), (2, 4), (5, 1), (13, 12)])
def test_frame_windows_cover_every_pair_once(n, window):
    seen = []
    for img, dep, m in sequence.FrameWindows(_frames(n), window):
        assert img.shape[0] == dep.shape[0] == m and 2 <= m <= window + 1
        ids = [int(img[k, 0, 0]) for k in range(m)]
        assert ids == [int(dep[k, 0, 0]) - 1000 for k in range(m)]
        assert ids == list(range(ids[0], ids[0] + m))
        seen += [(a, a + 1) for a in ids[:-1]]
    assert seen == [(a, a + 1) for a in range(n - 1)]          # consecutive windows share exactly one frame


@pytest.mark.parametrize("copy_threads", [0, 1, 4])
def test_frame_windows_copy_threads_and_ring_reuse(copy_threads):
    """Frames copied into the windows by a pool of threads land where the inline copy puts them; the ring of one FrameWindows can be
    handed to the next (page-locking a ring is a start-up cost), and a ring of another geometry is refused."""
    first = sequence.FrameWindows(_frames(11), 4, copy_threads=copy_threads)
    a = [(img.copy(), dep.copy(), m) for img, dep, m in first]
    assert len(first.buffers) == 3
    second = sequence.FrameWindows(_frames(11), 4, copy_threads=copy_threads, buffers=first.buffers)
    b = [(img.copy(), dep.copy(), m) for img, dep, m in second]
    assert second.buffers[0][0] is first.buffers[0][0]
    ref = [(img.copy(), dep.copy(), m) for img, dep, m in sequence.FrameWindows(_frames(11), 4, copy_threads=0)]
    for got in (a, b):
        assert len(got) == len(ref)
        for (i1, d1, m1), (i2, d2, m2) in zip(got, ref):
            assert m1 == m2 and np.array_equal(i1, i2) and np.array_equal(d1, d2)
    with pytest.raises(ValueError):
        sequence.FrameWindows(_frames(11), 5, buffers=first.buffers)


def test_align_windows_pipelines_through_submit_host():
    """An aligner with submit_host / drain (StreamedAligner) gets the windows pipelined: every window submitted once, results in
    sequence order, the last one drained; a window of another size arrives as it is (the aligner drains by itself)."""
    class Stub:
        def __init__(self):
            self.calls, self.inflight = [], None

        def _result(self, img):
            n = img.shape[0] - 1
            ids = img[:-1, 0, 0].astype(np.int32)
            return {"T": np.tile(np.eye(4, dtype=np.float32)[:3], (n, 1, 1)) * ids[:, None, None], "iters": ids, "err": ids.astype(np.float64)}

        def submit_host(self, img, dep, colour, max_distance):
            self.calls.append(("submit", img.shape[0]))
            prev, self.inflight = self.inflight, self._result(img)          # (computed at submit: the ring may recycle `img` later)
            return prev

        def drain(self):
            self.calls.append(("drain",))
            prev, self.inflight = self.inflight, None
            return prev

    al = Stub()
    res = sequence.align_windows(al, sequence.FrameWindows(_frames(11), 4), colour=False)
    assert al.calls == [("submit", 5), ("submit", 5), ("submit", 3), ("drain",)]
    assert list(res["iters"]) == list(range(10)) and res["T"].shape == (10, 3, 4)


def test_frame_windows_with_a_decoder_pool_keeps_the_order():
    import time

    def thunks(n):
        for k in range(n):
            def load(k=k):
                time.sleep(0.002 * ((k * 7) % 5))              # decoders finish out of order
                return np.full((6, 8), k, dtype=np.uint8), np.full((6, 8), 1000 + k, dtype=np.uint16)
            yield load

    ids = []
    for img, dep, m in sequence.FrameWindows(thunks(23), 5, workers=4):
        ids += [int(img[k, 0, 0]) for k in range(m if not ids else m)][(1 if ids else 0):]
    assert ids == list(range(23))


def test_frame_windows_colour_and_errors():
    wins = list(sequence.FrameWindows(_frames(5, colour=True), 2))
    assert wins[0][0].shape == (3, 6, 8, 3) and len(wins) == 2
    assert list(sequence.FrameWindows(_frames(1), 4)) == []    # one frame: no pair

    def bad():
        yield np.zeros((6, 8), np.uint8), np.zeros((6, 8), np.uint16)
        yield np.zeros((6, 9), np.uint8), np.zeros((6, 9), np.uint16)

    with pytest.raises(ValueError):
        list(sequence.FrameWindows(bad(), 4))

    def broken():
        yield np.zeros((6, 8), np.uint8), np.zeros((6, 8), np.uint16)
        raise OSError("unreadable png")

    with pytest.raises(OSError):                               # decoder errors surface in the consumer
        list(sequence.FrameWindows(broken(), 4))


def test_report_has_the_references_shape(tmp_path, golden_dir):
    """report.json keys / shapes of src/test_dvo.py:323-329 and the TUM txt of :337-345, from the chained reference run's own
    transforms: the 6-vector logs must reproduce the reference's recorded poses."""
    g = np.load(golden_dir / "sequence.npz")
    f = np.load(golden_dir / "frames.npz")
    T_rel = g["tf_T"][1:, :3, :]
    report, traj = sequence.build_report("TEST", T_rel, g["pose_T"][0].astype(np.float64), f["gt_T"], np.arange(10.0),
                                         {"input": "gray"})
    n = 10
    assert len(report["estimated_poses"]) == len(report["estimated_transforms"]) == len(report["errors"]) == n
    assert all(len(p) == 6 for p in report["estimated_poses"]) and report["estimated_transforms"][0] == [0.0] * 6
    np.testing.assert_allclose(np.array(report["estimated_poses"]), g["pose_xi"].reshape(n, 6), atol=2e-5)
    np.testing.assert_allclose(np.array(report["estimated_transforms"]), g["tf_xi"].reshape(n, 6), atol=2e-6)
    np.testing.assert_allclose(traj, g["pose_T"], atol=5e-6)
    assert report["errors"][0] < 1e-6 and all(isinstance(e, float) for e in report["errors"])
    assert 0.003 < report["ate_rmse_m"] < 0.006 and report["type"] == "TEST" and report["input"] == "gray"
    sequence.write_report(tmp_path, report, traj)
    back = json.loads((tmp_path / "report.json").read_text())
    assert back["estimated_poses"] == report["estimated_poses"]
    stamps, T = tum.read_tum_trajectory(tmp_path / "report.txt")
    np.testing.assert_allclose(stamps, np.arange(10.0))
    np.testing.assert_allclose(T, traj, atol=1e-5)          # (the float32 ground-truth start pose is orthonormal to ~3e-6 only)
    r2, _ = sequence.build_report("X", T_rel, None, None, np.arange(10.0))
    assert r2["errors"] == ["N/A"] * n and "ate_rmse_m" not in r2


def test_windows_and_shards_cover_any_sequence_exactly_once():
    """Property (hypothesis): for any sequence length, window size, ring depth, decoder-pool size and world size, the pairs seen by
    all ranks' windows are exactly the pairs of the sequence, each once, in order within a rank."""
    from hypothesis import given, settings, strategies as st
    from dvo_amd.batch import shard_pairs

    @settings(max_examples=60, deadline=None)
    @given(n=st.integers(2, 40), window=st.integers(1, 9), ring=st.integers(2, 4), workers=st.integers(0, 3),
           world=st.integers(1, 5))
    def check(n, window, ring, workers, world):
        seen = []
        for rank in range(world):
            lo, hi = shard_pairs(n - 1, rank, world)
            if hi <= lo:
                continue

            def frames(lo=lo, hi=hi):
                for k in range(lo, hi + 1):
                    item = (np.full((2, 3), k % 251, dtype=np.uint8), np.full((2, 3), k, dtype=np.uint16))
                    yield (lambda item=item: item) if k % 3 == 0 else item

            mine = []
            for img, dep, m in sequence.FrameWindows(frames(), window, ring=ring, workers=workers):
                ids = [int(dep[k, 0, 0]) for k in range(m)]
                assert ids == list(range(ids[0], ids[0] + m)) and all(int(img[k, 0, 0]) == ids[k] % 251 for k in range(m))
                mine += [(a, a + 1) for a in ids[:-1]]
            assert mine == [(a, a + 1) for a in range(lo, hi)]
            seen += mine
        assert seen == [(a, a + 1) for a in range(n - 1)]

    check()


def test_frame_windows_decoder_thread_ends_when_the_consumer_stops_early():
    import threading
    import time
    fw = sequence.FrameWindows(_frames(200), 2)
    it = iter(fw)
    next(it)
    it.close()                                    # consumer walks away after one window
    deadline = time.time() + 5.0
    while fw._thread.is_alive() and time.time() < deadline:
        time.sleep(0.05)
    assert not fw._thread.is_alive()
    assert threading.active_count() < 20


def _decode_stub(k, fail_at=-1):
    """picklable stand-in for a PNG decoder (module level: the pool's worker processes import it by name)"""
    if k == fail_at:
        raise OSError("cannot identify image file %d" % k)
    return np.full((4, 6, 3), k % 251, dtype=np.uint8), np.full((4, 6), k, dtype=np.uint16)


def test_decoder_pool_of_processes_keeps_order_and_surfaces_errors():
    """DecoderPool: decoders run in worker PROCESSES (started through the forkserver before anything touches a GPU), frames come back
    through shared memory in sequence order - also with fewer slots than frames in flight and with already-decoded items mixed in -
    and a failing decoder raises in the consumer at its place in the sequence."""
    import functools
    with sequence.DecoderPool(3, (4, 6, 3), (4, 6), slots=5) as pool:
        tasks = [functools.partial(_decode_stub, k) if k % 4 else _decode_stub(k) for k in range(57)]
        seen = []
        for img, dep, m in sequence.FrameWindows(tasks, 8, workers=pool):
            ids = [int(dep[k, 0, 0]) for k in range(m)]
            assert all(int(img[k, 1, 2, 0]) == ids[k] % 251 for k in range(m))
            seen += [(a, a + 1) for a in ids[:-1]]
        assert seen == [(a, a + 1) for a in range(56)]
        # a second sequence through the same pool, one decoder failing
        bad = [functools.partial(_decode_stub, k, 13) for k in range(30)]
        with pytest.raises(OSError, match="cannot identify image file 13"):
            list(sequence.FrameWindows(bad, 4, workers=pool))
    assert all(not p.is_alive() for p in pool._procs)

