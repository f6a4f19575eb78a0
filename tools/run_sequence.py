#!/usr/bin/env python3
"""Runs the MI355X direct-alignment core over an RGB-D sequence and reports the trajectory error.

    python tools/run_sequence.py --tum-dir /data/rgbd_dataset_freiburg1_desk [--size N]   # TUM RGB-D layout
    python tools/run_sequence.py --fixture                                                # the 10-frame fr1/xyz excerpt
    python tools/run_sequence.py --synthetic 64                                           # rendered stream with GT

Counterpart of the reference's CLI harness (src/test_dvo.py: loaders :86-280, main :298-345): same outputs
(report.json with estimated poses / transforms / errors, TUM-format trajectory txt) plus ATE / RPE, which the
reference does not compute.  Frame pairs are independent (init_guess = identity), so a window of frames is
aligned in one batched device-side solve.
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--tum-dir", type=Path)
    src.add_argument("--fixture", action="store_true")
    src.add_argument("--synthetic", type=int, metavar="N_FRAMES")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--weighting", default="none", choices=["none", "student_t_reference", "student_t", "huber"])
    ap.add_argument("--window", type=int, default=128, help="frames per batched solve")
    ap.add_argument("--camera", type=Path, default=None, help="camera_intrinsics.yaml (TUM mode; default fr1)")
    ap.add_argument("--out", type=Path, default=ROOT / "gpurun_out" / "sequence")
    ap.add_argument("--gpus", type=int, default=1, help="shard the pairs of the sequence over this many GPUs (one process each)")
    ap.add_argument("--spawn", action="store_true", help="go through the rank launcher even for --gpus 1")
    args = ap.parse_args()

    from dvo_amd import launch
    # BASELINE config 4: one process per GPU; `--gpus N` starts its own ranks (fresh children, before any HIP call in
    # this process) unless a launcher (torchrun) already did.  The pairs of the sequence are split into contiguous
    # ranges, one per rank; the pose records are all-gathered over RCCL and rank 0 composes / writes the trajectory.
    if not launch.started_by_launcher() and (args.gpus > 1 or args.spawn):
        try:
            raise SystemExit(launch.launch_ranks(args.gpus, str(Path(__file__).resolve()), [a for a in sys.argv[1:] if a != "--spawn"]))
        except launch.LaunchError as e:
            raise SystemExit("run_sequence.py: %s" % e)

    from dvo_amd import synthetic, tum, RGBDCameraModel
    from dvo_amd.batch import BatchAligner, compose_trajectory, shard_pairs, pose_records, allgather_poses

    import os
    rank, local_rank, world = launch.rank_env()
    use_dist = launch.started_by_launcher() or os.environ.get("DVO_FORCE_DIST") == "1"
    if use_dist:
        try:
            launch.check_world(args.gpus)
        except launch.LaunchError as e:
            raise SystemExit("run_sequence.py: %s" % e)
        import torch
        import torch.distributed as dist
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("run_sequence.py: rank %d has no GPU" % rank)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()

    gray_mode = True
    if args.fixture:
        f = np.load(ROOT / "tests" / "golden" / "frames.npz")
        grays, depths, gt = f["gray"], f["depth"], f["gt_T"]
        K, scale, stamps, kind = f["intrinsics"], float(f["depth_scale"]), np.arange(len(grays), dtype=float), "TEST"
        frames = list(zip(grays, depths))
    elif args.synthetic:
        st = synthetic.make_stream(args.synthetic, seed=0, device="cuda")
        frames = list(zip(st["gray"], st["depth"]))
        gt, K, scale, stamps, kind = st["T_wc"], st["K"], st["depth_scale"], np.arange(args.synthetic) / 30.0, "SYNTHETIC"
    else:
        seq = tum.load_tum_sequence(args.tum_dir, size=args.size)
        cam = RGBDCameraModel.load_from_yaml(args.camera) if args.camera else None
        K = cam.intrinsics[:3, :3] if cam else synthetic.TUM_FR1_K
        scale = cam.depth_scale if cam else synthetic.DEPTH_SCALE
        frames, gt, stamps, kind, gray_mode = tum.iter_frames(seq), seq.gt_T_wc, seq.timestamps, "TUM", False

    n_total = None
    if use_dist:
        frames = list(frames)                      # (a rank only needs its own range; kept simple here)
        n_total = len(frames)
        lo, hi = shard_pairs(n_total - 1, rank, world)
        frames = frames[lo:hi + 1]                 # pairs lo..hi-1 need frames lo..hi (one duplicated boundary frame)
    frames = iter(frames)
    first = next(frames)
    h, w = first[1].shape
    al = BatchAligner(K, scale, w, h, args.levels, max_pairs=args.window, max_iterations=args.iterations,
                      weighting=args.weighting, device=local_rank)
    T_rel, iters, t_solve, n_frames = [], [], 0.0, 1
    carry = first
    done = False
    while not done:
        batch = [carry]
        while len(batch) < args.window + 1:
            try:
                batch.append(next(frames))
            except StopIteration:
                done = True
                break
        if len(batch) < 2:
            break
        t0 = time.perf_counter()
        for i, (img, depth) in enumerate(batch):
            if gray_mode:
                al.ctx.upload_frame(i, img, depth, args.levels)
            else:
                al.ctx.upload_frame_bgr(i, img, depth, args.levels, 5.0)
        al.solve_enqueue(len(batch))
        res = al.solve_fetch(len(batch))
        t_solve += time.perf_counter() - t0
        T_rel.append(res["T"])
        iters.append(res["iters"])
        n_frames += len(batch) - 1
        carry = batch[-1]
    al.close()
    T_rel = np.concatenate(T_rel)
    if n_total is not None:
        import torch.distributed as dist
        counts = [shard_pairs(n_total - 1, r, world)[1] - shard_pairs(n_total - 1, r, world)[0] for r in range(world)]
        rec = allgather_poses(pose_records(T_rel, np.concatenate(iters)), counts, device="cuda")
        T_rel = rec[:, :12].reshape(-1, 3, 4)
        iters = [rec[:, 12]]
        n_frames = n_total
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return
    init = gt[0] if gt is not None else np.eye(4)
    traj = compose_trajectory(T_rel, init_pose=init)

    args.out.mkdir(parents=True, exist_ok=True)
    report = {"type": kind, "frames": int(n_frames), "seconds_upload_and_solve": t_solve,
              "frame_pairs_per_s_including_host_upload": (n_frames - 1) / t_solve,
              "gn_iterations_mean": float(np.concatenate(iters).mean()),
              "estimated_transforms": T_rel.reshape(len(T_rel), 12).tolist(),
              "estimated_poses": traj.reshape(len(traj), 16).tolist()}
    if gt is not None:
        report.update(tum.evaluate(traj, np.asarray(gt)[:len(traj)]))
        report["errors"] = np.linalg.norm(traj[:, :3, 3] - np.asarray(gt)[:len(traj), :3, 3], axis=1).tolist()
    with (args.out / "report.json").open("w") as fp:
        json.dump(report, fp, indent=1)
    tum.write_tum_trajectory(args.out / "report.txt", stamps[:len(traj)], traj)
    print(json.dumps({k: v for k, v in report.items() if not isinstance(v, list)}))
    if n_total is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
