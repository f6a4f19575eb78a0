#!/usr/bin/env python3
"""bench.py — frame-pairs/s of the full coarse-to-fine Gauss-Newton solve on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic RGB-D input per GPU: for a window of
B+1 consecutive VGA frames already resident in HBM, build the pyramids + packed Sobel planes on the
device, then run the whole 4-level x 10-iteration Gauss-Newton solve of the B frame pairs on the device
(BASELINE.json configs[1]: 640x480, 4-level pyramid, 10 GN iterations/level; synthetic stream of that
shape, since TUM fr1/desk is not on disk).  Each rank owns an independent stream (weak scaling);
the per-pair pose records are all-gathered over RCCL inside the timed region.

The JSON line also carries
  roofline     — the dominant kernel (level-0 k_reduce): algorithmic bytes of SURVEY 8d (2P + 10V per pair
                 per launch, with the measured V) / its average duration from HIP events on its own stream;
                 roofline.whole_step: the same figure for the whole job (SURVEY 8d bytes per frame pair x `value`);
                 roofline.valu_issue: what actually bounds that kernel - its instruction count (static, from the library that
                 was built; dynamic, the committed PMC reading) x the cost of a wave-instruction measured in this run by
                 dvo_debug_valu_rate, beside the measured launch time
  cpu_baseline — the NumPy oracle (a port of the reference CPU path) timed on this host on a bounded sample
and, at N = 1 on the default workload, outside the `value` region (BASELINE.json north_star's report):
  config3             — configs[2]: 1280x960, 5 levels, Huber: frame-pairs/s and its own level-0 roofline object
                        (also for N > 1: every rank runs it, rank 0 reports the aggregate and the per-rank figures)
  pcie_inclusive      — the same solve fed from page-locked host memory (gray and colour, 2 / 4 slices)
  student_t_reference — the reference's own robust weighting (use_weighter=True) on the metric's workload
  production_mode     — the build's Gauss-Newton mode (build-defined)
Everything under a key named `recorded_..._not_measured_in_this_run` (and `roofline.traffic`) is a committed counter
reading with the commit it was taken at (profiles/recorded_counters.json, profiles/pmc_traffic.json), not a product of this run.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
PMC_FILE = ROOT / "profiles" / "pmc_traffic.json"   # HBM bytes per launch from rocprofv3 --pmc passes (tools/profile_gpu.sh)


def recorded_traffic(kernel_pairs, width, height):
    """HBM bytes per level-0 launch measured with rocprofv3 PMC counters (separate FETCH_SIZE / WRITE_SIZE passes,
    gfx950 x2 read correction) for the same launch shape, or None.  PMC counters cannot be read from inside this
    process; the value is the committed measurement of the same command (profiles/)."""
    try:
        rec = json.loads(PMC_FILE.read_text())
        for e in rec["entries"]:
            if e["pairs_per_launch"] == kernel_pairs and e["width"] == width and e["height"] == height:
                return e
    except Exception:
        pass
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default: ~1.2 s of GPU work at 1024 VGA pairs per step)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="frame pairs per GPU per step")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--weighting", default="none", choices=["none", "student_t_reference", "student_t", "huber"])
    ap.add_argument("--variant", default="v1", choices=["v1", "v2", "v3", "gn"],
                    help="v1: reference CPU default (the metric); v2: approximate_image2_gradient; v3: numba.cuda kernel semantics")
    ap.add_argument("--level-iterations", type=str, default=None, help="comma-separated iterations per level, finest first (build-defined schedule; default: --iterations at every level)")
    ap.add_argument("--streams", type=int, default=4, help="independent HIP streams (sub-batches) per GPU")
    ap.add_argument("--distinct", type=int, default=0, help="distinct rendered frames (0 = batch+1)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through the launcher even for --gpus 1 (rehearses the multi-GPU entry on one GPU)")
    ap.add_argument("--config3-all-ranks", action="store_true",
                    help="run the 1280x960 leg on every rank and gather it (default for N > 1; this flag forces it for the one-rank rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs beside the metric (production mode, Student-t, PCIe-inclusive, 1280x960 config 3, fixture ATE): A/B scripts")
    ap.add_argument("--cpu-pairs", type=int, default=16, help="frame pairs of the CPU baseline sample (~0.35 s each on one thread)")
    return ap.parse_args()


def _median_iteration_ms(O, g0, d0, g1, K, depth_scale, n_iter):
    """One level-0 Gauss-Newton iteration of the oracle = the reference's call graph per iteration (deproject -> J_w -> transform ->
    project -> 3x bilinear -> J -> float32 J^T J -> 6x6 solve): median of n_iter runs after one warm-up."""
    import numpy as np
    lv = O.PairLevel(g0, d0, g1, O.level_intrinsics(K, 0), depth_scale)
    T = O.Pose.from_xi(np.array([0.002, -0.001, 0.003, 0.001, -0.002, 0.0015])).matrix()
    ts = []
    for k in range(n_iter + 1):
        t0 = time.perf_counter()
        r, J, _, _ = O.residuals_and_jacobian(lv, T)
        H, b, _, _ = O.normal_equations(r, J, None, "fp32")
        O.solve_increment(H, b)
        if k:
            ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def cpu_baseline(stream_np, levels, iterations, n_pairs):
    """the NumPy oracle (a port of the reference's CPU path, float32 J^T J like the reference) on this host, on a
    bounded sample of the bench workload: per level-0 GN iteration (median of 20) and per full frame pair, with one thread and
    with every core NumPy's BLAS / a worker pool may use, plus a 1280x960 leg.  The real reference adds numba-parallel loops
    and ~25 ms of np_cache key hashing per iteration (utils/numpy_cache.py:33-43): this is a favourable stand-in for it."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "oracle"))
    import dvo_oracle as O
    from threadpoolctl import threadpool_limits, threadpool_info
    g, d = stream_np["gray"], stream_np["depth"]
    K, sc = stream_np["K"], stream_np["depth_scale"]
    n_pairs = min(n_pairs, g.shape[0] - 1)
    host_cpus = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:                                             # pragma: no cover
        usable = host_cpus
    # --- one thread ---------------------------------------------------------------------------------------------------
    with threadpool_limits(limits=1):
        it_ms_1 = _median_iteration_ms(O, g[0], d[0], g[1], K, sc, 20)
        O.gauss_newton(g[0], d[0], g[1], K, sc, levels=levels, max_iterations=1, accumulate="fp32")     # warm-up
        t0 = time.perf_counter()
        for k in range(n_pairs):
            O.gauss_newton(g[k], d[k], g[k + 1], K, sc, levels=levels, max_iterations=iterations, accumulate="fp32")
        dt = time.perf_counter() - t0
    out = {"value": n_pairs / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
           "sample": "%d VGA frame pairs x %d levels x %d GN iterations, NumPy oracle (float32 J^T J), 1 thread, %.1f s; host has %d "
                     "logical CPUs (%d usable by this process)" % (n_pairs, levels, iterations, dt, host_cpus, usable),
           "per_iteration_ms_vga_level0": {"threads_1": it_ms_1, "iterations_timed": 20},
           "host_logical_cpus": host_cpus, "usable_cpus": usable}
    # --- every BLAS thread the process may use (the only parallelism inside one NumPy iteration is the J^T J sgemm) ------
    try:
        blas_threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
        out["per_iteration_ms_vga_level0"]["threads_all"] = _median_iteration_ms(O, g[0], d[0], g[1], K, sc, 20)
        out["per_iteration_ms_vga_level0"]["blas_threads"] = blas_threads
    except Exception as e:                                        # pragma: no cover
        out["per_iteration_ms_vga_level0"]["threads_all_error"] = repr(e)
    # --- every core of ONE GPU's share of the node: independent pairs over worker processes (pairs are independent, as on the GPU).
    # An MI355X node carries 8 GPUs, so the share is usable_cpus / 8 (32 of 256 on this pool), capped at 64; each worker aligns one
    # pair of the sample or more.  Plain child processes running oracle/cpu_worker.py: NumPy + oracle only, never torch / HIP.  Each
    # reports the seconds of its own range (start-up and a warm-up pair excluded); they start together.
    try:
        import subprocess
        import tempfile
        workers = max(1, min(64, usable // 8 if usable >= 8 else usable))
        n_all = min(max(n_pairs, workers), g.shape[0] - 1)              # at least one pair per worker
        workers = min(workers, n_all)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "cpu_sample.npz")
            np.savez(path, gray=g[:n_all + 1], depth=d[:n_all + 1], K=K, depth_scale=sc)
            bounds = [round(i * n_all / workers) for i in range(workers + 1)]
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            procs = [subprocess.Popen([sys.executable, str(ROOT / "oracle" / "cpu_worker.py"), path, str(bounds[i]), str(bounds[i + 1]),
                                       str(levels), str(iterations)], stdout=subprocess.PIPE, env=env, text=True) for i in range(workers)]
            secs = [float(p.communicate(timeout=600)[0].strip().splitlines()[-1]) for p in procs]
        dtp = max(secs)
        out["all_workers"] = {"value": n_all / dtp, "unit": "frame-pairs/s", "cores": workers,
                              "sample": "%d pairs of the same stream over %d single-threaded worker processes (one GPU's share of the node's %d usable "
                                        "CPUs: 1/8), slowest worker %.1f s" % (n_all, workers, usable, dtp)}
    except Exception as e:                                            # pragma: no cover
        out["all_workers"] = {"error": repr(e)}
    # --- 1280x960, 5 levels (BASELINE config 3 shape): the VGA frames upsampled 2x (same texture statistics), one thread ------
    try:
        big = [np.repeat(np.repeat(a, 2, axis=0), 2, axis=1) for a in (g[0], d[0], g[1])]
        Kb = np.array(K, dtype=np.float32).copy()
        Kb[:2, :] *= 2.0
        with threadpool_limits(limits=1):
            it_big = _median_iteration_ms(O, big[0], big[1], big[2], Kb, sc, 5)
            t0 = time.perf_counter()
            O.gauss_newton(big[0], big[1], big[2], Kb, sc, levels=5, max_iterations=iterations, accumulate="fp32")
            dtb = time.perf_counter() - t0
        out["w1280x960"] = {"per_iteration_ms_level0_threads_1": it_big, "iterations_timed": 5,
                            "frame_pairs_per_s_threads_1": 1.0 / dtb,
                            "sample": "one 1280x960 pair (VGA frames upsampled 2x), 5 levels x %d iterations, 1 thread, %.1f s" % (iterations, dtb)}
    except Exception as e:                                            # pragma: no cover
        out["w1280x960"] = {"error": repr(e)}
    return out


RECORDED_FILE = ROOT / "profiles" / "recorded_counters.json"   # counter readings taken OUTSIDE this run (rocprofv3 PMC), with their commit


def recorded_counters():
    """Counter-derived facts about the dominant kernel that cannot be measured inside this process (rocprofv3 PMC passes; accuracy
    runs): read from profiles/recorded_counters.json together with the commit / date they were taken at.  Reported under keys
    that say so; nothing in them is produced by this run."""
    try:
        return json.loads(RECORDED_FILE.read_text())
    except Exception:
        return None


def roofline_leg(part, gray_ptr, depth_ptr, n_pairs, steps):
    """Every reduction launch of `steps` full solves on ONE stream bracketed by HIP events on that stream (each timed launch has
    the GPU to itself, like a rocprofv3 kernel trace) -> (launches per level, ms per level)."""
    part.ctx.profile_enable(True)
    part.ctx.profile_read(reset=True)
    for _ in range(steps):
        part.ingest_device(gray_ptr, depth_ptr, n_pairs + 1)
        part.solve_enqueue(n_pairs + 1)
        part.solve_fetch(n_pairs + 1)
    launches, ms = part.ctx.profile_read(reset=True)
    part.ctx.profile_enable(False)
    return launches, ms


def timed_steps(fn, sync, steps, warmup=1):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    sync()
    return (time.perf_counter() - t0) / steps, out


def timed_windows(al, d_gray, d_depth, n_frames, sync, steps, warmup=1):
    """`steps` windows through StreamedAligner.submit_device / drain (a slice's next window is enqueued when its previous result
    has been fetched: no drain of the GPU between steps), every result fetched inside the timed region.  -> (s per step, last result)"""
    for _ in range(warmup):
        al.submit_device(d_gray, d_depth, n_frames)
    al.drain()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        al.submit_device(d_gray, d_depth, n_frames)
    out = al.drain()
    sync()
    return (time.perf_counter() - t0) / steps, out


def leg_student_t(st, gray, depth, B, W, H, L, iterations, streams, device, steps):
    """The reference's own robust mode (use_weighter=True: TDistributionWeighter, weighter/t_weighter.py:21-47) on the metric's
    workload: same stream, same batch, device solve."""
    import torch
    from dvo_amd.batch import StreamedAligner
    al = StreamedAligner(st["K"], st["depth_scale"], W, H, L, max_pairs=B, n_streams=streams, device=device,
                         max_iterations=iterations, weighting="student_t_reference")
    try:
        def sync():
            al.synchronize()
            torch.cuda.synchronize()
        dt, res = timed_windows(al, gray.data_ptr(), depth.data_ptr(), B + 1, sync, steps)
    finally:
        al.close()
    return {"frame_pairs_per_s": B / dt, "ms_per_step": 1e3 * dt, "steps": steps, "batch_pairs": B,
            "weighting": "student_t_reference (TDistributionWeighter as written, summing quirk included)",
            "gn_iterations_per_pair": float(res["iters"].sum()) / B}


def leg_pcie(st, gray, depth, B, W, H, L, iterations, device, steps):
    """PCIe-inclusive rate (never `value`): the same frames start in page-locked HOST memory; a step = batched upload + pyramid
    build + full solve, slices of the batch overlapping their copies with each other's kernels (dvo_set_upload_peer)."""
    import numpy as np
    import torch
    from dvo_amd import _cabi
    from dvo_amd.batch import StreamedAligner
    n = B + 1
    gp = _cabi.pinned_array((n, H, W), np.uint8)
    dp = _cabi.pinned_array((n, H, W), np.uint16)
    gp[...] = gray[:n].cpu().numpy()
    dp[...] = depth[:n].cpu().numpy().view(np.uint16)
    bgr = _cabi.pinned_array((n, H, W, 3), np.uint8)
    bgr[...] = gp[:, :, :, None]
    out = {"batch_pairs": B, "host_memory": "page-locked", "steps": steps}
    for slices in (2, 4):
        al = StreamedAligner(st["K"], st["depth_scale"], W, H, L, max_pairs=B, n_streams=slices, device=device, max_iterations=iterations)
        try:
            def sync():
                al.synchronize()
                torch.cuda.synchronize()
            dt, _ = timed_steps(lambda: al.align_host(gp, dp), sync, steps)
            out["gray_%d_slices" % slices] = {"frame_pairs_per_s": B / dt, "link_GBs": n * W * H * 3 / dt / 1e9}
            dt, _ = timed_steps(lambda: al.align_host_bgr(bgr, dp), sync, steps)
            out["colour_%d_slices" % slices] = {"frame_pairs_per_s": B / dt, "link_GBs": n * W * H * 5 / dt / 1e9}
        finally:
            al.close()
    del gp, dp, bgr
    return out


def leg_config3(device, iterations, streams, steps, seed=1234):
    """BASELINE.json configs[2]: synthetic 1280x960 RGB-D stream, 5-level pyramid, Huber weighting, 1 GPU.  The largest batch of
    1024 / 512 / 256 / 128 pairs that fits (1024 pairs in 4 streams = 256 pairs per launch, the size from which a launch reaches its
    rate; 1029 frames are ~30 GB of the 288); 129 rendered frames played forwards and backwards (every pair is two consecutive renders)."""
    import numpy as np
    import torch
    from dvo_amd import synthetic
    from dvo_amd.batch import StreamedAligner
    W, H, L = 1280, 960, 5
    n_render = 129
    st = synthetic.make_stream(n_render, W, H, seed=seed, device="cuda", as_torch=True)
    last_err = None
    for B in (1024, 512, 256, 128):
        al = None
        try:
            idx = np.arange(B + 1) % (2 * (n_render - 1))
            idx = np.where(idx < n_render, idx, 2 * (n_render - 1) - idx)              # 0..128,127..0,1..: consecutive renders only
            sel = torch.as_tensor(idx, device="cuda")
            gray = st["gray"].index_select(0, sel).contiguous()
            depth = st["depth"].index_select(0, sel).contiguous()
            al = StreamedAligner(st["K"], st["depth_scale"], W, H, L, max_pairs=B, n_streams=streams, device=device,
                                 max_iterations=iterations, weighting="huber")

            def sync():
                al.synchronize()
                torch.cuda.synchronize()
            dt, res = timed_windows(al, gray.data_ptr(), depth.data_ptr(), B + 1, sync, steps)
            part = al.parts[0]
            npr = min(B, part.max_pairs)
            launches, ms = roofline_leg(part, gray.data_ptr(), depth.data_ptr(), npr, 2)
            valid_l0 = float((depth[:npr] != 0).sum().item()) / npr
            alg = (2.0 * W * H + 10.0 * valid_l0) * npr
            avg_ms = ms[0] / max(launches[0], 1)
            out = {"workload": "synthetic 1280x960 RGB-D stream, 5-level pyramid, %d GN iterations/level, Huber weighting (build-defined: "
                               "the reference has no Huber), %d frame pairs per step in %d streams, ingest + device solve timed" % (iterations, B, streams),
                   "frame_pairs_per_s": B / dt, "ms_per_step": 1e3 * dt, "steps": steps, "batch_pairs": B,
                   "gn_iterations_per_pair": float(res["iters"].sum()) / B,
                   "roofline": {"kernel": "k_reduce<V1,huber> level 0 (1280x960), %d pairs per launch" % npr, "avg_launch_us": 1e3 * avg_ms,
                                "launches_timed": int(launches[0]), "algorithmic_bytes_per_launch": alg,
                                "achieved": alg / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "valid_fraction_level0": valid_l0 / (W * H)}}
            al.close()
            return out
        except Exception as e:                                   # out of memory at this batch: try the next smaller one
            last_err = repr(e)
            if al is not None:
                try:
                    al.close()
                except Exception:
                    pass
            torch.cuda.empty_cache()
    return {"error": last_err}


def leg_sequence_png(device):
    """SURVEY 8(f) f2 where it binds: a TUM-layout dataset of PNG files on disk (480 frames listed over 48 distinct renders) through
    the sequence front end - PNG decode in a thread pool into page-locked windows, batched colour upload + ingest, device solve -
    with 8 and 16 decoder threads, and the same windows fed from memory (tools/exp_sequence_png.py; the reference harness decodes
    the whole sequence into RAM first and times step() on decoded frames, src/test_dvo.py:180-191, :305-309).  Bounded to a few
    seconds; the full sweep is profiles/r04_sequence_png.txt."""
    sys.path.insert(0, str(ROOT / "tools"))
    import exp_sequence_png
    t0 = time.perf_counter()
    out = exp_sequence_png.measure(n_frames=480, n_distinct=48, threads=(8, 16), window=96, streams=2, device=device, colour=True,
                                   decode_only=False, from_memory=True)
    out["leg_seconds"] = time.perf_counter() - t0
    return out


def fixture_ate(device):
    """"ATE vs ref" part of the metric on the only real TUM data in the repo: the reference's 10-frame fr1/xyz
    excerpt (tests/golden/frames.npz) through the HIP path, next to the ATE of the reference run's own trajectory
    (tests/golden/sequence.npz, produced by oracle/refrun/make_golden.py)."""
    import numpy as np
    from dvo_amd.batch import BatchAligner, compose_trajectory, absolute_trajectory_error
    try:
        f = np.load(ROOT / "tests" / "golden" / "frames.npz")
        ref = np.load(ROOT / "tests" / "golden" / "sequence.npz")
    except Exception:
        return None
    al = BatchAligner(f["intrinsics"], float(f["depth_scale"]), 640, 480, 4, max_pairs=9, device=device, max_iterations=10)
    res = al.align_host(f["gray"], f["depth"])
    al.close()
    traj = compose_trajectory(res["T"], init_pose=f["gt_T"][0])
    return {"sequence": "TUM fr1/xyz, first 10 frames (reference tests/test_data), 4 levels x 10 iterations",
            "ate_rmse_m": absolute_trajectory_error(traj, f["gt_T"]),
            "reference_ate_rmse_m": absolute_trajectory_error(ref["pose_T"].astype(np.float64), f["gt_T"])}


class _HipBackend:
    """What main() needs from the machine: the real thing.  A CPU rehearsal of the distributed branch swaps in tests/bench_stub.py
    through DVO_BENCH_STUB (a stand-in aligner on gloo; no GPU, no kernels - it exists so that the N > 1 code of THIS file runs
    somewhere before an 8-GPU node shows up)."""
    name = "hip"
    device = "cuda"
    dist_backend = "nccl"

    def check(self, local_rank, rank):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no GPU (local rank %d, %d device(s) visible)" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def init_process_group(self, local_rank, timeout):
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=timeout)

    def make_stream(self, n, W, H, seed):
        from dvo_amd import synthetic
        return synthetic.make_stream(n, W, H, seed=seed, device="cuda", as_torch=True)

    def aligner(self, *a, **kw):
        from dvo_amd.batch import StreamedAligner
        return StreamedAligner(*a, **kw)

    def roofline_leg(self, part, gray_ptr, depth_ptr, n_pairs, steps):
        return roofline_leg(part, gray_ptr, depth_ptr, n_pairs, steps)

    def leg_config3(self, device, iterations, streams, steps, seed=1234):
        return leg_config3(device, iterations, streams, steps, seed=seed)

    def n_simds(self, local_rank):
        import torch
        return 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count


def _load_backend():
    path = os.environ.get("DVO_BENCH_STUB")
    if not path:
        return _HipBackend()
    import importlib.util
    spec = importlib.util.spec_from_file_location("dvo_bench_stub", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Backend()


ISA_FILE = ROOT / "profiles" / "kernel_isa.json"       # static facts + hash of the level-0 kernel's instructions (tools/isa_count.py --write)


def valu_issue_leg(ctx, n_simds, point_waves_per_launch, measured_us, rec):
    """The bound the level-0 launch actually runs at: VALU issue.  Instruction counts: static, from the library loaded in
    THIS process (tools/isa_count.py: disassembly of the kernel, its record loop), and dynamic, the committed PMC reading
    (SQ_INSTS_VALU per point-wave) - used only if the kernel's instruction hash still equals the one the reading was taken on.  Cost of
    a wave-instruction: dvo_debug_valu_rate, measured now, at the kernel's 5 waves per SIMD.  predicted = count x cost; beside it the
    measured launch."""
    out = {}
    try:
        sys.path.insert(0, str(ROOT / "tools"))
        import isa_count
        from dvo_amd import _cabi
        facts = isa_count.kernel_facts(_cabi.library_path())
        committed = json.loads(ISA_FILE.read_text()) if ISA_FILE.exists() else {}
        same = bool(committed) and committed.get("sha256") == facts["sha256"]
        out.update({"kernel_sha256": facts["sha256"], "committed_sha256": committed.get("sha256"), "isa_matches_committed": same,
                    "instr_per_point_static": facts["loop"]["valu_per_point"], "static_by_class_per_point": facts["loop"]["by_class_per_point"]})
    except Exception as e:                                                  # no llvm-objdump on the box: the probe still runs
        same = False
        out["isa_error"] = repr(e)
    lim = (rec or {}).get("level0_reduction") or {}
    recorded_ok = same and lim.get("kernel_sha256") == out.get("kernel_sha256")
    dyn = lim.get("valu_instructions_per_point_wave")
    out["instr_per_point_recorded"] = dyn
    out["recorded_reading_is_of_this_kernel"] = bool(recorded_ok)
    rate = ctx.valu_rate(waves_per_simd=5, iterations=4000)                # ~15 ms per composition
    out["ns_per_wave_instr"] = rate["mix"]
    out["ns_per_wave_instr_by_class"] = {k: rate[k] for k in ("fast", "conflict", "slow", "rcp")}
    # the order of the classes matters: the same composition in small interleaved blocks, and the sum of the class prices (no
    # interaction at all) - the band the headline figure sits in
    out["ns_per_wave_instr_other_orders"] = {"small_interleaved_blocks": rate["mix_small_blocks"], "class_priced_sum": rate["mix_class_priced"]}
    comp = rate["composition"]
    out["probe_composition"] = dict(zip(("fast", "fast_bank_conflict", "slow", "trans"), comp))
    # the probe issues the composition of one trip of the record loop as it was when the probe was last updated: same as the kernel's now?
    out["probe_composition_matches_kernel"] = ("kernel_sha256" in out) and all(
        facts["loop"]["by_class"].get(k, 0) == v for k, v in out["probe_composition"].items())
    out["probe"] = ("dvo_debug_valu_rate: ALU-only launch issuing the composition of one trip of the record loop (%d simple, %d simple with "
                    "both multiplicands in one VGPR bank, %d conversion / bit-field / compare / select / shift, %d v_rcp_f32) arranged as the "
                    "compiled loop arranges it (per point: unpack cluster, then the FMAs), 5 waves per SIMD, this run" % tuple(comp))
    ipp = dyn if (recorded_ok and dyn) else out.get("instr_per_point_static")
    out["instr_per_point"] = ipp
    out["instr_per_point_source"] = "recorded PMC reading (SQ_INSTS_VALU / point-waves)" if (recorded_ok and dyn) else "static count of the record loop (tools/isa_count.py)"
    if ipp:
        pred = point_waves_per_launch * ipp * rate["mix"] / n_simds * 1e-3
        out.update({"point_waves_per_launch": point_waves_per_launch, "simds": n_simds, "predicted_us": pred, "measured_us": measured_us,
                    "frac_of_issue_bound": pred / measured_us if measured_us else None,
                    "frac_of_issue_bound_other_orders": {k: (point_waves_per_launch * ipp * v / n_simds * 1e-3) / measured_us
                                                         for k, v in out["ns_per_wave_instr_other_orders"].items()} if measured_us else None})
    return out


def level_valid_fractions(part, L, sample=4):
    """Depth-valid fraction of every pyramid level, measured on the first `sample` frames resident in the first slice's slots."""
    fr = []
    for lvl in range(L):
        v = []
        for slot in range(sample):
            d = part.ctx.download_level(slot, lvl, want=("depth",))["depth"]
            v.append(float((d != 0).mean()))
        fr.append(sum(v) / len(v))
    return fr


def survey_bytes_per_pair(W, H, L, valid_fraction, iters_per_level):
    """per GN iteration at a level 2 P + 10 V; per frame pair once: pyramid 3 (P_{l-1} + P_l) for l >= 1 and gradients 9 P_l."""
    P = []
    w, h = W, H
    for _ in range(L):
        P.append(w * h)
        w, h = (w + 1) // 2, (h + 1) // 2
    it = sum(iters_per_level[l] * (2.0 * P[l] + 10.0 * valid_fraction[l] * P[l]) for l in range(L))
    setup = sum(3.0 * (P[l - 1] + P[l]) for l in range(1, L)) + 9.0 * sum(P)
    return it + setup, it, setup


def main():
    args = parse_args()
    from dvo_amd import launch
    hw = _load_backend()

    # `python bench.py --gpus N` starts its own N ranks (one process per GPU) when no launcher did: fresh child
    # processes under torch.distributed.run, spawned before this process has made any HIP call; rank 0's JSON line is
    # relayed through the inherited stdout.  Fewer than N visible devices is an error, never a 1-GPU measurement.
    if not launch.started_by_launcher() and (args.gpus > 1 or args.spawn):
        try:
            rc = launch.launch_ranks(args.gpus, str(Path(__file__).resolve()), [a for a in sys.argv[1:] if a != "--spawn"],
                                     need_devices=(hw.name == "hip"))
        except launch.LaunchError as e:
            raise SystemExit("bench.py: %s" % e)
        raise SystemExit(rc)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank, local_rank, world = launch.rank_env()
    launch.mark_rank("started")
    try:
        launch.check_world(args.gpus)
    except launch.LaunchError as e:
        raise SystemExit("bench.py: %s" % e)
    hw.check(local_rank, rank)
    dev = hw.device
    # under a launcher the process group is always formed (also for one rank: the same RCCL code path as N ranks)
    use_dist = launch.started_by_launcher() or os.environ.get("DVO_BENCH_FORCE_DIST") == "1"
    # (the process group is formed further down, AFTER this rank's device buffers and streams exist: creating the RCCL
    # communicator first costs the whole step ~4 % on MI355X — measured, profiles/r02_experiments.txt)

    from dvo_amd.batch import pose_records, PoseGatherer

    B, W, H, L = args.batch, args.width, args.height, args.levels
    n_frames = B + 1
    n_distinct = args.distinct if args.distinct > 0 else n_frames
    # independent stream per rank (BASELINE config 5), rendered on the device, resident in HBM before timing
    st = hw.make_stream(min(n_distinct, n_frames), W, H, 1000 + rank)
    gray, depth = st["gray"], st["depth"]
    if gray.shape[0] < n_frames:                      # optional: tile a shorter render (documented in config)
        reps = (n_frames + gray.shape[0] - 1) // gray.shape[0]
        gray = gray.repeat(reps, 1, 1)[:n_frames].contiguous()
        depth = depth.repeat(reps, 1, 1)[:n_frames].contiguous()
    hw.sync()
    valid_l0 = float((depth[:-1] != 0).sum().item()) / B            # measured V per pair at level 0

    al = hw.aligner(st["K"], st["depth_scale"], W, H, L, max_pairs=B, n_streams=args.streams, device=local_rank,
                    max_iterations=args.iterations, weighting=args.weighting,
                    approximate_image2_gradient=(args.variant == "v2"),
                    reference_variant={"v3": "cuda", "gn": "gn"}.get(args.variant, "cpu"),
                    level_iterations=([int(v) for v in args.level_iterations.split(",")] if args.level_iterations else None))
    launch.phase("aligner built (%d pairs in %d streams)" % (B, args.streams))
    if use_dist:
        hw.init_process_group(local_rank, launch.pg_timeout())
        world = dist.get_world_size()                      # the ranks that actually joined
        if world != args.gpus:
            raise SystemExit("bench.py: %d ranks joined the process group, --gpus %d" % (world, args.gpus))
        launch.phase("process group formed (world size %d, backend %s)" % (world, dist.get_backend()))
    gather = PoseGatherer([B] * world, device=dev) if use_dist else None

    gather_s = [0.0]

    def share(res):
        if use_dist and res is not None:
            tg = time.perf_counter()
            allrec = gather(pose_records(res["T"], res["iters"], res["err"]))       # every rank ends up with all poses of the step
            gather_s[0] += time.perf_counter() - tg
            assert allrec.shape[0] == B * world
        return res

    # A step = ingest + solve of one window of B pairs.  The windows are pipelined (StreamedAligner.submit_device): each of the
    # `streams` slices of a window is enqueued as soon as the slice's previous result has been fetched, so the GPU does not drain
    # between steps; step() returns (and, for N > 1, all-gathers) the result of the PREVIOUS step, finish() that of the last one:
    # K steps = K windows solved, K results fetched and K gathers, all inside the timed region.
    def step():
        return share(al.submit_device(gray.data_ptr(), depth.data_ptr(), n_frames))

    def finish():
        return share(al.drain())

    for _ in range(args.warmup):
        step()
    res = finish()
    if res is not None and not np.all(np.isfinite(res["xi"])):
        raise SystemExit("non-finite poses in warm-up")
    launch.phase("warm-up done")

    def fence():
        al.synchronize()
        hw.sync()
        if use_dist:
            dist.barrier()

    fence()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    res = finish()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_rank = elapsed
    launch.phase("timed loop done (%d steps, %.3f ms per step on this rank)" % (args.steps, 1e3 * elapsed / args.steps))
    # the same steps with a fetch of every slice at the end of each (the GPU drains between steps): reported beside `value`
    sync_steps = max(1, min(args.steps, 10))
    ts0 = time.perf_counter()
    for _ in range(sync_steps):
        al.align_device(gray.data_ptr(), depth.data_ptr(), n_frames)
    fence()
    step_synchronous_ms = 1e3 * (time.perf_counter() - ts0) / sync_steps
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    iters_total = int(res["iters"].sum())

    # --- roofline leg: same steps again with every reduction launch bracketed by HIP events ----------------
    # (run on ONE stream so that each timed launch has the GPU to itself, like the rocprofv3 kernel trace; every rank does it,
    # so that a multi-GPU record shows each rank's kernel time next to its step time)
    prof = al.parts[0]
    npr = min(B, prof.max_pairs)
    launches, ms = hw.roofline_leg(prof, gray.data_ptr(), depth.data_ptr(), npr, max(1, min(args.steps, 5)))
    launch.phase("roofline leg done")
    per_rank = None
    if use_dist:
        mine = torch.tensor([1e3 * elapsed_rank / args.steps, 1e3 * ms[0] / max(launches[0], 1)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        per_rank = {"ms_per_step": [float(v) for v in allr[:, 0]], "level0_avg_launch_us": [float(v) for v in allr[:, 1]],
                    "ms_per_step_min": float(allr[:, 0].min()), "ms_per_step_max": float(allr[:, 0].max())}

    # --- the build's own production mode beside the metric (N = 1, default workload only): true Gauss-Newton steps on the same
    # kernel (DVO_VARIANT_GN), 1 iteration at the finest level and 2 at the others.  Reported under "production_mode"; never `value`.
    gn_mode = None
    default_workload = (args.variant == "v1" and args.weighting == "none" and not args.level_iterations and (W, H, L) == (640, 480, 4))
    extras = (world == 1 and rank == 0 and not args.no_extras and default_workload and hw.name == "hip")
    rec = recorded_counters() or {}
    whole_step = valu = None
    if rank == 0 and hw.name == "hip":
        # SURVEY 8(d) bytes of the whole job: per level the measured valid fraction and the iterations that actually ran
        try:
            vf = level_valid_fractions(prof, L)
            sched = [int(v) for v in args.level_iterations.split(",")] if args.level_iterations else [args.iterations] * L
            sched = (sched + [sched[-1]] * L)[:L]
            nominal = float(sum(sched))
            scale_it = (iters_total / B) / nominal if nominal > 0 else 1.0       # pairs that stop early run fewer iterations
            bpp, b_it, b_setup = survey_bytes_per_pair(W, H, L, vf, [scale_it * v for v in sched])
            rate = world * B * args.steps / elapsed
            whole_step = {"algorithmic_bytes_per_pair": bpp, "iteration_bytes_per_pair": b_it, "setup_bytes_per_pair": b_setup,
                          "valid_fraction_per_level": vf, "gn_iterations_per_pair": iters_total / B,
                          "achieved": bpp * rate / world / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": bpp * rate / world / 1e9 / HBM_PEAK_GBS,
                          "note": "SURVEY 8(d) bytes per frame pair (2 P + 10 V per iteration and level, pyramid + gradients once) x frame-pairs/s per GPU"}
        except Exception as e:                                         # pragma: no cover
            whole_step = {"error": repr(e)}
        if default_workload or (W, H) == (640, 480):
            try:
                valu = valu_issue_leg(prof.ctx, hw.n_simds(local_rank), valid_l0 * npr / 64.0, 1e3 * ms[0] / max(launches[0], 1), rec)
            except Exception as e:                                     # pragma: no cover
                valu = {"error": repr(e)}
    if extras:
        from dvo_amd import _cabi
        sched = [1] + [2] * (L - 1)                       # finest level first (tools/exp_gn_schedule.sh)
        for a in al.parts:
            a.variant, a.max_iterations = _cabi.VARIANT_GN, max(sched)
            a.ctx.set_level_iterations(sched)
        gn_steps = max(1, min(args.steps, 10))
        step()
        finish()
        fence()
        tg0 = time.perf_counter()
        for _ in range(gn_steps):
            step()
        res_gn = finish()
        fence()
        tg = time.perf_counter() - tg0
        gn_mode = {"variant": "gn (DVO_VARIANT_GN: V1 with the Sobel response normalised and J_w at the warped point; build-defined, "
                              "not a reference behaviour)", "gn_iterations_per_level_finest_first": sched, "frame_pairs_per_s": B * gn_steps / tg,
                   "ms_per_step": 1e3 * tg / gn_steps, "steps": gn_steps,
                   "recorded_accuracy_not_measured_in_this_run": rec.get("production_mode_accuracy")}
        for a in al.parts:
            a.variant, a.max_iterations = _cabi.VARIANT_V1, args.iterations
            a.ctx.set_level_iterations(None)

    if rank == 0:
        P0 = W * H
        alg_bytes = (2.0 * P0 + 10.0 * valid_l0) * npr                # SURVEY 8d, per level-0 launch
        layout_bytes = (4.0 * P0 + 8.0 * valid_l0) * npr              # this build's layout: 8-byte valid-point records of
                                                                      # frame 1 + the 4-byte packed texels of frame 2
        avg_ms = ms[0] / max(launches[0], 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        per_level = {str(l): {"launches": int(launches[l]), "avg_us": (1e3 * ms[l] / launches[l]) if launches[l] else None}
                     for l in range(L)}
        pairs_per_s = world * B * args.steps / elapsed
        traffic = recorded_traffic(npr, W, H) or {}
        # a counter reading describes the kernel it was taken on: stale when the instruction hash of the kernel built here differs
        sha_now = (valu or {}).get("kernel_sha256")
        lim = rec.get("level0_reduction")
        if lim is not None:
            lim = dict(lim, stale=(sha_now is None or lim.get("kernel_sha256") != sha_now))
        out = {
            "metric": "frame-pairs/s (full pyramid GN solve) VGA RGB-D",
            "value": pairs_per_s,
            "unit": "frame-pairs/s",
            "n_gpus": world, "ranks_joined": (dist.get_world_size() if use_dist else 1),
            "collective": ("%s all-gather of pose records, world size %d" % ("RCCL" if hw.name == "hip" else dist.get_backend(), dist.get_world_size())) if use_dist else "none (single process)",
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "float32 per-pixel arithmetic (as the reference), float64 accumulation and 6x6 solve",
            "data": "synthetic",
            "config": {"workload": "synthetic %dx%d RGB-D stream, %d-level pyramid, %d GN iterations/level, "
                                   "%d frame pairs per GPU per step (pyramid + Sobel + valid-point-list build and device-side GN solve "
                                   "inside the timed region), weighting=%s, variant=%s%s" % (W, H, L, args.iterations, B, args.weighting, args.variant,
                                                                                          (", iterations per level (finest first) " + args.level_iterations) if args.level_iterations else ""),
                       "batch_pairs_per_gpu": B, "streams_per_gpu": args.streams, "distinct_frames": int(min(n_distinct, n_frames)),
                       "stepping": "pipelined: a slice's next window is enqueued when its previous result has been fetched; every result "
                                   "of the K steps is fetched inside the timed region",
                       "step_synchronous": {"ms_per_step": step_synchronous_ms, "frame_pairs_per_s": world * B / (1e-3 * step_synchronous_ms),
                                            "steps": sync_steps, "note": "this rank's time with all slices fetched at the end of every step"},
                       "gn_iterations_per_pair": iters_total / B,
                       "gn_iterations_per_s": world * iters_total * args.steps / elapsed,
                       "pose_gather_ms_per_step": (1e3 * gather_s[0] / args.steps) if use_dist else None,
                       "parallelism": "1 independent stream per GPU, RCCL all-gather of pose records"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # `traffic`: HBM bytes per launch from rocprofv3 PMC counters.  PMC counters cannot be read from inside this
                         # process: the value is the committed reading of the same launch shape (profiles/pmc_traffic.json), taken
                         # at `traffic_recorded_at`; null when no reading of this shape exists; `traffic_stale` when the kernel built
                         # here is not the one it was taken on
                         "traffic": traffic.get("hbm_bytes_per_launch"),
                         "traffic_recorded_at": traffic.get("commit"),
                         "traffic_source": traffic.get("source"),
                         "traffic_stale": (traffic.get("hbm_bytes_per_launch") is not None) and (sha_now is None or traffic.get("kernel_sha256") != sha_now),
                         "kernel": "k_reduce<V1,%s> level 0 (%dx%d), %d pairs per launch" % (args.weighting, W, H, npr),
                         "avg_launch_us": 1e3 * avg_ms, "launches_timed": int(launches[0]),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "layout_bytes_per_launch": layout_bytes,
                         "achieved_layout_GBs": layout_bytes / (avg_ms * 1e-3) / 1e9,
                         "valid_fraction_level0": valid_l0 / P0, "per_level": per_level,
                         "whole_step": whole_step,
                         "valu_issue": valu,
                         "recorded_limiter_not_measured_in_this_run": lim},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if gn_mode is not None:
            out["production_mode"] = gn_mode
    al.close()
    al = None
    # config 3 (1280x960 x 5 levels, Huber) on EVERY rank when N > 1: north_star asks for both image sizes at 1 / 2 / 4 / 8 GPUs.  Each
    # rank runs its own independent stream (no data-path collective); rank 0 reports the aggregate and the per-rank figures.
    c3_multi = None
    if use_dist and ((world > 1 and not args.no_extras) or args.config3_all_ranks) and default_workload:
        try:
            c3 = hw.leg_config3(local_rank, args.iterations, args.streams, max(2, min(args.steps, 10)), seed=1234 + rank)
            mine = [c3["frame_pairs_per_s"], c3["roofline"]["frac"], c3["roofline"]["avg_launch_us"], float(c3["batch_pairs"])]
        except Exception as e:                                             # a rank whose leg fails still joins the collective
            print("rank %d: config 3 leg failed: %r" % (rank, e), file=sys.stderr, flush=True)
            c3, mine = None, [float("nan")] * 4
        launch.phase("config 3 done" if c3 is not None else "config 3 FAILED")
        tm = torch.tensor(mine, dtype=torch.float64, device=dev)
        allc = [torch.zeros_like(tm) for _ in range(world)]
        dist.all_gather(allc, tm)
        allc = torch.stack(allc).cpu().numpy()
        if rank == 0:
            c3_multi = dict(c3 or {"error": "rank 0 leg failed"})
            failed = [int(r) for r in np.flatnonzero(~np.isfinite(allc[:, 0]))]
            c3_multi.update({"n_gpus": world, "frame_pairs_per_s_per_rank": [float(v) if np.isfinite(v) else None for v in allc[:, 0]],
                             "frame_pairs_per_s": float(np.nansum(allc[:, 0])) if not failed else None,
                             "failed_ranks": failed,
                             "roofline_frac_per_rank": [float(v) if np.isfinite(v) else None for v in allc[:, 1]],
                             "level0_avg_launch_us_per_rank": [float(v) if np.isfinite(v) else None for v in allc[:, 2]],
                             "aggregate_note": "sum over ranks of independent streams run concurrently (weak scaling); `roofline` is rank 0's; "
                                               "null when a rank's leg failed (failed_ranks)"})
    if rank == 0:
        if c3_multi is not None:
            out["config3"] = c3_multi
        if extras:
            # the rest of what north_star asks to be reported, on the driver's clock, outside the `value` region and bounded in time
            # (each leg builds and frees its own contexts; a leg that fails reports its error instead of taking the line down)
            es = max(2, min(args.steps, 10))
            for name, fn in (("student_t_reference", lambda: leg_student_t(st, gray, depth, B, W, H, L, args.iterations, args.streams, local_rank, es)),
                             ("pcie_inclusive", lambda: leg_pcie(st, gray, depth, min(B, 512), W, H, L, args.iterations, local_rank, max(2, min(es, 5)))),
                             ("config3", lambda: leg_config3(local_rank, args.iterations, args.streams, es)),
                             ("sequence_png", lambda: leg_sequence_png(local_rank))):
                try:
                    out[name] = fn()
                except Exception as e:                                         # pragma: no cover
                    out[name] = {"error": repr(e)}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_extras and hw.name == "hip":
            out["accuracy"] = fixture_ate(local_rank)
        if not args.no_cpu_baseline and world == 1 and hw.name == "hip":            # CPU baseline: rank 0 at N=1 only
            cpu_frames = min(max(args.cpu_pairs, 64) + 1, n_frames)            # (the all-workers leg wants a pair per worker)
            st_np = {"gray": gray[:cpu_frames].cpu().numpy(),
                     "depth": depth[:cpu_frames].cpu().numpy().view(np.uint16),
                     "K": st["K"], "depth_scale": st["depth_scale"]}
            out["cpu_baseline"] = cpu_baseline(st_np, L, args.iterations, args.cpu_pairs)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    launch.mark_rank("done")


if __name__ == "__main__":
    main()
