"""Multi-rank path on CPU (gloo, world_size 2): pair sharding, pose-record all-gather, pose chaining.
The GPU ranks use the same code with backend "nccl" (= RCCL over xGMI)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from dvo_amd.batch import shard_pairs, pose_records, compose_trajectory, absolute_trajectory_error

ROOT = Path(__file__).resolve().parents[1]


def test_shard_pairs_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 100, 797):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_pairs(n, r, world)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
            sizes = [shard_pairs(n, r, world)[1] - shard_pairs(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_pairs(4, 2, 2)


def test_compose_trajectory_matches_reference_sequence(golden_dir):
    """pose_t = pose_{t-1} * T_t^-1 (core/base_dense_visual_odometry.py:79) against the chained reference run."""
    g = np.load(golden_dir / "sequence.npz")
    poses = compose_trajectory(g["tf_T"][1:, :3, :], init_pose=g["pose_T"][0].astype(np.float64))
    np.testing.assert_allclose(poses, g["pose_T"], atol=5e-6)
    rec = pose_records(g["tf_T"][1:, :3, :], iters=np.full(9, 40), err=np.ones(9))
    np.testing.assert_allclose(compose_trajectory(rec, init_pose=g["pose_T"][0]), poses, atol=1e-6)


def test_ate_is_invariant_to_rigid_motion():
    rng = np.random.default_rng(0)
    gt = np.tile(np.eye(4), (20, 1, 1))
    gt[:, :3, 3] = np.cumsum(rng.normal(0, 0.05, (20, 3)), axis=0)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    est = gt.copy()
    est[:, :3, 3] = gt[:, :3, 3] @ R.T + np.array([1.0, -2.0, 0.5])
    assert absolute_trajectory_error(est, gt) < 1e-12
    est[7, :3, 3] += np.array([0.1, 0, 0])
    assert 0.005 < absolute_trajectory_error(est, gt) < 0.05


_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["DVO_PKG_DIR"])
from dvo_amd.batch import shard_pairs, pose_records, allgather_poses, compose_trajectory, PoseGatherer

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = np.load(os.environ["DVO_GOLDEN"])
T_all = g["tf_T"][1:, :3, :]                       # nine relative transforms of the reference run
n = T_all.shape[0]
lo, hi = shard_pairs(n, rank, world)
counts = [shard_pairs(n, r, world)[1] - shard_pairs(n, r, world)[0] for r in range(world)]
local = pose_records(T_all[lo:hi], iters=np.arange(lo, hi), err=np.full(hi - lo, float(rank)))
allrec = allgather_poses(local, counts)
assert allrec.shape == (n, 16), allrec.shape
assert np.array_equal(allrec[:, :12].reshape(n, 3, 4), T_all.astype(np.float32))
assert np.array_equal(allrec[:, 12], np.arange(n, dtype=np.float32))            # order preserved across ranks
assert np.array_equal(allrec[:, 13], np.repeat(np.arange(world, dtype=np.float32), counts))
gather = PoseGatherer(counts)                      # the step-loop form: persistent buffers, same result, call after call
for _ in range(3):
    assert np.array_equal(gather(local), allrec)
poses = compose_trajectory(allrec, init_pose=g["pose_T"][0].astype(np.float64))
assert np.abs(poses - g["pose_T"]).max() < 5e-6
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_poses_gloo(tmp_path, golden_dir, world):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, DVO_PKG_DIR=str(ROOT / "dense-visual-odometry_amd"),
               DVO_GOLDEN=str(golden_dir / "sequence.npz"), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for r in range(world):
        assert "rank %d ok" % r in out.stdout


_SHARD_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["DVO_PKG_DIR"])
from dvo_amd import sequence
from dvo_amd.batch import shard_pairs

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
f = np.load(os.environ["DVO_FRAMES"])
g = np.load(os.environ["DVO_GOLDEN"])
grays, depths = f["gray"], f["depth"]
n_total = grays.shape[0]
key = {int(grays[k].astype(np.int64).sum()): k for k in range(n_total)}           # frame content -> frame index
calls = []


class FakeAligner:
    """Stands in for the GPU aligner: recognises the frames of a window by content and returns the reference run's own
    transforms for their pairs, so the test pins the data flow (ranges, windows, carry frames, gather order), not the kernels."""
    def align_host(self, gray, depth):
        idx = [key[int(gray[k].astype(np.int64).sum())] for k in range(gray.shape[0])]
        assert idx == list(range(idx[0], idx[0] + len(idx))), idx
        assert all(np.array_equal(depth[k], depths[i]) for k, i in enumerate(idx))
        calls.append(idx)
        T = np.stack([g["tf_T"][i + 1][:3, :] for i in idx[:-1]]).astype(np.float32)
        return {"T": T, "iters": np.array(idx[:-1], dtype=np.int32), "err": np.zeros(len(idx) - 1)}
    def close(self):
        calls.append("closed")


def source(lo, hi):
    for k in range(lo, hi):
        yield (lambda k=k: (grays[k], depths[k])) if k % 2 else (grays[k], depths[k])   # decoder thunks and plain frames mixed

res = sequence.align_shard(lambda h, w, win: FakeAligner(), source, n_total, rank, world, window_pairs=2, colour=False, workers=2)
lo, hi = shard_pairs(n_total - 1, rank, world)
assert (res["lo"], res["hi"]) == (lo, hi) and res["T"].shape[0] == hi - lo and calls[-1] == "closed"
assert [c[0] for c in calls[:-1]] == list(range(lo, hi, 2))                      # windows of 2 pairs, sharing their boundary frame
T_all, iters_all = sequence.gather_shards(res["T"], res["iters"], n_total)
assert np.array_equal(iters_all, np.arange(n_total - 1, dtype=np.float32))       # pair order preserved across ranks
report, traj = sequence.build_report("TEST", T_all, g["pose_T"][0].astype(np.float64), f["gt_T"], np.arange(float(n_total)))
assert np.abs(traj - g["pose_T"]).max() < 5e-6
assert abs(report["ate_rmse_m"] - 4.1317e-3) < 1e-6
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_sequence_flow_gloo(tmp_path, golden_dir, world):
    """BASELINE config 4 on CPU: every rank aligns its contiguous range of pairs in windows (decoder pool, carry frames), the pose
    records are all-gathered (gloo here, RCCL on the GPUs) and the composed trajectory equals the reference run's."""
    script = tmp_path / "shard_worker.py"
    script.write_text(_SHARD_WORKER)
    env = dict(os.environ, DVO_PKG_DIR=str(ROOT / "dense-visual-odometry_amd"), DVO_GOLDEN=str(golden_dir / "sequence.npz"),
               DVO_FRAMES=str(golden_dir / "frames.npz"), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    for r in range(world):
        assert "rank %d ok" % r in out.stdout


def _run_bench_stub(tmp_path, extra_env=None, timeout=300):
    """`bench.py --gpus 2` under torch.distributed.run on CPU: bench.py's OWN distributed branch with the stand-in backend of
    tests/bench_stub.py (gloo, world size 2)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, DVO_BENCH_STUB=str(root / "tests" / "bench_stub.py"), OMP_NUM_THREADS="1", DVO_PG_TIMEOUT="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2",
           str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "8", "--streams", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(root))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                              # rank 0 alone prints
    return json.loads(lines[0]), out.stderr


def test_bench_distributed_branch_world_size_two_on_gloo(tmp_path):
    """bench.py's own N > 1 code: process group formed after the aligner, one pose gather per step inside the timed region, the time
    of the SLOWEST rank, per-rank step and launch times, config 3 on every rank with its aggregate - and the phase lines a killed
    SCALE run would leave on stderr."""
    d, err = _run_bench_stub(tmp_path)
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and "world size 2" in d["collective"]
    assert d["scaling"] == "weak" and d["steps"] == 4
    pr = d["per_rank"]
    # (the per-step gather keeps the ranks in lock step, so both report about the slower rank's time: rank 1 sleeps 4 ms per step)
    assert len(pr["ms_per_step"]) == 2 and min(pr["ms_per_step"]) >= 3.5
    assert d["ms_per_step"] >= pr["ms_per_step_max"] * (1 - 1e-9)                                    # value is priced on the slowest rank
    assert abs(d["value"] - 2 * 8 / (1e-3 * d["ms_per_step"])) < 1e-6 * d["value"]                   # pairs of ALL ranks / that time
    assert pr["level0_avg_launch_us"][1] > pr["level0_avg_launch_us"][0]
    assert d["config"]["pose_gather_ms_per_step"] is not None and d["config"]["pose_gather_ms_per_step"] >= 0
    c3 = d["config3"]
    assert c3["n_gpus"] == 2 and c3["frame_pairs_per_s_per_rank"] == [1000.0, 2000.0] and c3["frame_pairs_per_s"] == 3000.0
    assert c3["failed_ranks"] == [] and np.allclose(c3["roofline_frac_per_rank"], [0.4, 0.41])
    for k in (0, 1):
        for what in ("aligner built", "process group formed (world size 2", "timed loop done", "config 3 done"):
            assert ("rank %d: %s" % (k, what)) in err, (k, what, err[-2000:])


def test_bench_distributed_branch_reports_a_failed_config3_rank(tmp_path):
    """A rank whose config-3 leg raises still joins the gather (NaN record): rank 0 names it, gives no aggregate, and the run ends
    normally instead of hanging the other ranks in the collective."""
    d, err = _run_bench_stub(tmp_path, {"DVO_BENCH_STUB_FAIL_C3_RANK": "1"})
    c3 = d["config3"]
    assert c3["failed_ranks"] == [1] and c3["frame_pairs_per_s"] is None
    assert c3["frame_pairs_per_s_per_rank"] == [1000.0, None] and c3["roofline_frac_per_rank"][1] is None
    assert "rank 1: config 3 FAILED" in err and d["value"] > 0

