"""Stand-in backend for bench.py (DVO_BENCH_STUB): lets the DISTRIBUTED branch of bench.py itself - process group, per-step pose gather
inside the timed region, max-over-ranks time, per-rank figures, the config-3 gather incl. the failed-rank path - run on CPU with gloo at
world size 2, long before an 8-GPU node does.  No kernels, no oracle: the aligner returns fixed poses after a short sleep.  Test
infrastructure only; the product path never imports it."""
import os
import time

import numpy as np


class _Part:
    max_pairs = 0
    ctx = None


class _StubAligner:
    def __init__(self, K, depth_scale, W, H, L, max_pairs=64, n_streams=2, device=0, **kw):
        self.max_pairs = max_pairs
        self.parts = [_Part() for _ in range(n_streams)]
        for p in self.parts:
            p.max_pairs = (max_pairs + n_streams - 1) // n_streams
        self._inflight = None
        self._iters = int(kw.get("max_iterations", 10)) * L
        self._rank = int(os.environ.get("RANK", "0"))

    def _result(self, n_frames):
        n = n_frames - 1
        T = np.tile(np.eye(4, dtype=np.float32)[:3], (n, 1, 1))
        T[:, 0, 3] = 0.01 * (self._rank + 1)
        return {"T": T, "xi": np.zeros((n, 6)), "iters": np.full(n, self._iters, dtype=np.int32), "err": np.ones(n)}

    def submit_device(self, gray_ptr, depth_ptr, n_frames):
        time.sleep(0.002 * (1 + self._rank))                  # rank 1 is the slower one: the max over ranks must show it
        prev, self._inflight = self._inflight, n_frames
        return self._result(prev) if prev is not None else None

    def drain(self):
        prev, self._inflight = self._inflight, None
        return self._result(prev) if prev is not None else None

    def align_device(self, gray_ptr, depth_ptr, n_frames):
        time.sleep(0.002 * (1 + self._rank))
        return self._result(n_frames)

    def synchronize(self):
        pass

    def close(self):
        pass


class Backend:
    name = "stub"
    device = "cpu"
    dist_backend = "gloo"

    def check(self, local_rank, rank):
        pass

    def sync(self):
        pass

    def init_process_group(self, local_rank, timeout):
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", timeout=timeout)

    def make_stream(self, n, W, H, seed):
        import torch
        g = torch.zeros((n, 8, 8), dtype=torch.uint8)
        d = torch.ones((n, 8, 8), dtype=torch.int16)
        return {"gray": g, "depth": d, "K": np.eye(3, dtype=np.float32), "depth_scale": 0.0002}

    def aligner(self, *a, **kw):
        return _StubAligner(*a, **kw)

    def roofline_leg(self, part, gray_ptr, depth_ptr, n_pairs, steps):
        rank = int(os.environ.get("RANK", "0"))
        launches = np.zeros(8, dtype=np.int64)
        ms = np.zeros(8)
        launches[0], ms[0] = 10 * steps, 10 * steps * 0.175 * (1 + 0.1 * rank)
        return launches, ms

    def leg_config3(self, device, iterations, streams, steps, seed=1234):
        rank = int(os.environ.get("RANK", "0"))
        if os.environ.get("DVO_BENCH_STUB_FAIL_C3_RANK") == str(rank):
            raise RuntimeError("stub: config 3 leg failed on rank %d" % rank)
        return {"workload": "stub 1280x960", "frame_pairs_per_s": 1000.0 * (rank + 1), "ms_per_step": 1.0, "steps": steps, "batch_pairs": 256,
                "roofline": {"frac": 0.4 + 0.01 * rank, "avg_launch_us": 380.0 + rank}}

    def n_simds(self, local_rank):
        return 1024
