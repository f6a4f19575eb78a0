"""One process per GPU: starts the ranks of a multi-GPU run of `bench.py` / `tools/run_sequence.py`.

`python bench.py --gpus N` must produce an N-rank measurement by itself (the driver's command shape), so when a
script is asked for N > 1 GPUs and was not itself started by a launcher (no WORLD_SIZE in the environment) it calls
`launch_ranks` *before anything has touched the GPU*: N fresh child processes under `torch.distributed.run`
(rendezvous on 127.0.0.1, a free port), stdout / stderr relayed, exit code returned.  Nothing is ever exec'ed from a
process that has initialised HIP: the children are new interpreters, and this parent only counts devices
(`torch.cuda.device_count()` does not initialise the runtime on this image) and waits.

The reference has no distributed code; this is the build's own contract for BASELINE configs 4-5.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Optional, Sequence


class LaunchError(RuntimeError):
    pass


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def started_by_launcher(env=None) -> bool:
    """True inside a rank started by torch.distributed.run / torchrun (or any launcher that sets the usual variables)."""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def visible_devices() -> int:
    """Number of HIP devices, without initialising the runtime in this process."""
    import torch
    return int(torch.cuda.device_count())


def rank_env(env=None):
    """(rank, local_rank, world_size) of this process; (0, 0, 1) when not under a launcher."""
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def check_world(requested_gpus: int, env=None) -> int:
    """Inside a rank: the process group size the launcher gave us must be the number of GPUs that was asked for —
    a mismatch is an error, never a silent single-GPU measurement."""
    _, _, world = rank_env(env)
    if world != requested_gpus:
        raise LaunchError("--gpus {} but the launcher started WORLD_SIZE={} ranks".format(requested_gpus, world))
    return world


def launch_ranks(n_ranks: int, script: str, argv: Sequence[str], need_devices: bool = True,
                 extra_env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Runs `script argv...` as n_ranks processes (one per GPU) under torch.distributed.run and returns the exit code.
    The children's stdout / stderr are inherited, so rank 0's JSON line is this process's output too.
    need_devices: fail (LaunchError) when fewer than n_ranks HIP devices are visible instead of degrading."""
    if n_ranks < 1:
        raise LaunchError("n_ranks must be >= 1")
    if need_devices:
        have = visible_devices()
        if have < n_ranks:
            raise LaunchError("{} GPUs requested but only {} HIP device(s) visible".format(n_ranks, have))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    if extra_env:
        env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
        raise LaunchError("ranks did not finish within {} s".format(timeout))
