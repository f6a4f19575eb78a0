#!/usr/bin/env python3
"""Runs the MI355X direct-alignment core over an RGB-D sequence and reports the trajectory error.

    python tools/run_sequence.py --tum-dir /data/rgbd_dataset_freiburg1_desk [--size N]   # TUM RGB-D layout
    python tools/run_sequence.py --fixture                                                # the 10-frame fr1/xyz excerpt
    python tools/run_sequence.py --synthetic 64 [--colour]                                # rendered stream with GT
    python tools/run_sequence.py --synthetic 4096 --gpus 8                                # BASELINE config 4 shape

Counterpart of the reference's CLI harness (src/test_dvo.py: loaders :86-280, main :298-345): same outputs
(report.json with the reference's keys — 6-vector logs of poses / transforms, per-frame errors — and a TUM-format
trajectory txt) plus the 4x4 forms and ATE / RPE, which the reference does not compute.  Frame pairs are independent
(init_guess = identity), so frames are decoded by a background thread into page-locked windows, each window goes up
with one batched upload and all its pairs are aligned in one device-side solve (dvo_amd/sequence.py).
`--gpus N` starts N ranks (one process per GPU), each aligning a contiguous range of pairs; the pose records are
all-gathered over RCCL and rank 0 composes and writes the trajectory.
"""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))

import numpy as np  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--tum-dir", type=Path)
    src.add_argument("--fixture", action="store_true")
    src.add_argument("--synthetic", type=int, metavar="N_FRAMES")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--weighting", default="none", choices=["none", "student_t_reference", "student_t", "huber"])
    ap.add_argument("--variant", default="v1", choices=["v1", "v2", "v3", "gn"],
                    help="v1 / v2 / v3: the reference's behaviours (CPU default, approximate_image2_gradient, numba.cuda kernel); "
                         "gn: the build's true Gauss-Newton mode (DVO_VARIANT_GN), which needs ~3 iterations per level")
    ap.add_argument("--level-iterations", type=str, default=None, help="comma-separated iterations per level, finest first (overrides --iterations)")
    ap.add_argument("--window", type=int, default=128, help="frame pairs per batched upload + solve")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams (slices of a window) per GPU")
    ap.add_argument("--decoders", type=int, default=16, help="PNG decoder threads (TUM mode; PIL stops scaling at ~16: profiles/r04_sequence_png.txt)")
    ap.add_argument("--decoder-processes", type=int, default=0, help="decode in this many worker PROCESSES instead of threads (TUM mode): "
                                                                     "scales with the cores; the pool is started before the GPU is touched")
    ap.add_argument("--colour", action="store_true", help="feed BGR frames (conversion + depth clamp on the device) "
                                                           "also for --fixture / --synthetic (B = G = R)")
    ap.add_argument("--camera", type=Path, default=None, help="camera_intrinsics.yaml (TUM mode; default fr1)")
    ap.add_argument("--out", type=Path, default=ROOT / "gpurun_out" / "sequence")
    ap.add_argument("--reference-cli-compat", action="store_true",
                    help="reproduce the data flow of the reference's own CLI for like-for-like comparisons with its runs: colour frames "
                         "reach the BGR2GRAY ingest in RGB order (src/test_dvo.py:183 + core/base_dense_visual_odometry.py:58: red / blue "
                         "weights swapped) and the TUM trajectory is stamped with the associated GROUND-TRUTH timestamps (:203, :337-345) "
                         "instead of the frame stamps.  With --fixture: the two colour fixture frames of tests/golden/cli_compat.npz")
    ap.add_argument("--gpus", type=int, default=1, help="shard the pairs of the sequence over this many GPUs (one process each)")
    ap.add_argument("--spawn", action="store_true", help="go through the rank launcher even for --gpus 1")
    return ap.parse_args()


def main():
    args = parse_args()
    from dvo_amd import launch
    # BASELINE config 4: one process per GPU; `--gpus N` starts its own ranks (fresh children, before any HIP call in
    # this process) unless a launcher (torchrun) already did.
    if not launch.started_by_launcher() and (args.gpus > 1 or args.spawn):
        try:
            raise SystemExit(launch.launch_ranks(args.gpus, str(Path(__file__).resolve()), [a for a in sys.argv[1:] if a != "--spawn"]))
        except launch.LaunchError as e:
            raise SystemExit("run_sequence.py: %s" % e)

    from dvo_amd import synthetic, tum, sequence, _cabi, RGBDCameraModel
    from dvo_amd.batch import StreamedAligner

    rank, local_rank, world = launch.rank_env()
    launch.mark_rank("started")
    # decoder worker processes are started BEFORE anything in this process initialises HIP (a process that has must not exec)
    pool = None
    if args.tum_dir and args.decoder_processes > 0:
        seq0 = tum.load_tum_sequence(args.tum_dir, size=args.size)
        f0 = tum.read_frame(seq0.rgb_paths[0], seq0.depth_paths[0], "bgr")
        pool = sequence.DecoderPool(args.decoder_processes, f0[0].shape, f0[1].shape)
    use_dist = launch.started_by_launcher() or os.environ.get("DVO_FORCE_DIST") == "1"
    if use_dist:
        try:
            launch.check_world(args.gpus)
        except launch.LaunchError as e:
            raise SystemExit("run_sequence.py: %s" % e)
        import torch
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("run_sequence.py: rank %d has no GPU" % rank)
        torch.cuda.set_device(local_rank)
        # (the process group is formed after this rank's alignment: creating the RCCL communicator before the device buffers
        # and streams exist slows the kernels by ~4 %, profiles/r02_experiments.txt; rank and world size come from the launcher)

    # -- the sequence: (n_total, frame source for a range, ground truth, intrinsics, stamps) ------------------------
    colour = args.colour
    compat = args.reference_cli_compat
    if compat and args.synthetic:
        raise SystemExit("run_sequence.py: --reference-cli-compat applies to colour images on disk (--tum-dir, --fixture)")
    if args.fixture and compat:
        # the reference CLI's own data flow on the fixture's colour frames 1, 2 (oracle/refrun/make_golden.py: cli_compat_case)
        c = np.load(ROOT / "tests" / "golden" / "cli_compat.npz")
        f = np.load(ROOT / "tests" / "golden" / "frames.npz")
        rgbs, depths, gt = [c["rgb_1"], c["rgb_2"]], [c["depth_1"], c["depth_2"]], f["gt_T"][:2]
        K, scale, stamps, kind, colour, n_total = f["intrinsics"], float(f["depth_scale"]), np.arange(2, dtype=float), "TEST", True, 2

        def source(lo, hi):
            for k in range(lo, hi):
                yield rgbs[k], depths[k]                 # RGB order into the BGR ingest, as the reference CLI does
    elif args.fixture:
        f = np.load(ROOT / "tests" / "golden" / "frames.npz")
        grays, depths, gt = f["gray"], f["depth"], f["gt_T"]
        K, scale, stamps, kind = f["intrinsics"], float(f["depth_scale"]), np.arange(len(grays), dtype=float), "TEST"
        n_total = len(grays)

        def source(lo, hi):
            for k in range(lo, hi):
                yield (np.repeat(grays[k][..., None], 3, axis=2) if colour else grays[k]), depths[k]
    elif args.synthetic:
        st = synthetic.make_stream(args.synthetic, seed=0, device="cuda")
        grays, depths = st["gray"], st["depth"]
        gt, K, scale, stamps, kind = st["T_wc"], st["K"], st["depth_scale"], np.arange(args.synthetic) / 30.0, "SYNTHETIC"
        n_total = args.synthetic

        def source(lo, hi):
            for k in range(lo, hi):
                yield (np.repeat(grays[k][..., None], 3, axis=2) if colour else grays[k]), depths[k]
    else:
        seq = tum.load_tum_sequence(args.tum_dir, size=args.size)
        cam = RGBDCameraModel.load_from_yaml(args.camera) if args.camera else None
        K = cam.intrinsics[:3, :3] if cam else synthetic.TUM_FR1_K
        scale = cam.depth_scale if cam else synthetic.DEPTH_SCALE
        gt, stamps, kind, colour, n_total = seq.gt_T_wc, seq.timestamps, "TUM", True, len(seq)
        if compat and seq.gt_timestamps is not None:
            stamps = seq.gt_timestamps               # src/test_dvo.py:203: the txt carries the ground-truth stamps
        order = "rgb" if compat else "bgr"

        def source(lo, hi):                          # decoded lazily, by the decoder pool, for this rank's range only
            import functools
            for k in range(lo, hi):
                yield functools.partial(tum.read_frame, seq.rgb_paths[k], seq.depth_paths[k], order)      # (picklable: worker processes)

    def make_aligner(h, w, window):
        return StreamedAligner(K, scale, w, h, args.levels, max_pairs=window, n_streams=args.streams, device=local_rank,
                               max_iterations=args.iterations, weighting=args.weighting,
                               approximate_image2_gradient=(args.variant == "v2"),
                               reference_variant={"v3": "cuda", "gn": "gn"}.get(args.variant, "cpu"),
                               level_iterations=([int(v) for v in args.level_iterations.split(",")] if args.level_iterations else None))

    res = sequence.align_shard(make_aligner, source, n_total, rank if use_dist else 0, world if use_dist else 1, args.window,
                               colour, alloc=_cabi.pinned_array, workers=(pool if pool is not None else args.decoders), streams=args.streams)
    if pool is not None:
        pool.close()
    n_local, decode_s = res["hi"] - res["lo"], res["decode_seconds"]
    T_rel, iters = res["T"], res["iters"]

    if use_dist:
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=launch.pg_timeout())
        if dist.get_world_size() != world:
            raise SystemExit("run_sequence.py: %d ranks joined, %d expected" % (dist.get_world_size(), world))
        T_rel, iters = sequence.gather_shards(T_rel, iters, n_total, device="cuda")
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            launch.mark_rank("done")
            return

    extra = {"frames": int(n_total), "n_gpus": int(world), "seconds_decode_upload_and_solve": res["seconds"],
             "seconds_decode_thread": decode_s,
             "frame_pairs_per_s_including_decode_and_upload": (n_local / res["seconds"]) * world if res["seconds"] > 0 else None,
             "gn_iterations_mean": float(np.mean(iters)) if len(iters) else 0.0,
             "input": ("rgb fed as bgr (reference CLI)" if compat else "bgr") if colour else "gray",
             "variant": args.variant, "max_iterations_per_level": args.iterations, "level_iterations": args.level_iterations,
             "reference_cli_compat": bool(compat)}
    report, traj = sequence.build_report(kind, T_rel, gt[0] if gt is not None else None, gt, stamps, extra)
    sequence.write_report(args.out, report, traj)
    print(json.dumps({k: v for k, v in report.items() if not isinstance(v, list)}))
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()
    launch.mark_rank("done")


if __name__ == "__main__":
    main()
