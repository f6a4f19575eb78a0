"""One process per GPU: starts the ranks of a multi-GPU run of `bench.py` / `tools/run_sequence.py`.

`python bench.py --gpus N` must produce an N-rank measurement by itself (the driver's command shape), so when a
script is asked for N > 1 GPUs and was not itself started by a launcher (no WORLD_SIZE in the environment) it calls
`launch_ranks` *before anything has touched the GPU*: N fresh child processes under `torch.distributed.run`
(`--standalone --local-addr 127.0.0.1`: the rendezvous store picks its own port), stdout / stderr inherited, exit code
returned; a finite timeout after which the whole process group of the launcher is terminated.  Nothing is ever exec'ed from a
process that has initialised HIP: the children are new interpreters, and this parent only counts devices
(`torch.cuda.device_count()` does not initialise the runtime on this image) and waits.

The reference has no distributed code; this is the build's own contract for BASELINE configs 4-5.
"""
from __future__ import annotations

import os
import subprocess
import sys
from typing import Optional, Sequence


class LaunchError(RuntimeError):
    pass


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


# A rank stuck in rendezvous / RCCL init must not hang the command until the caller's own limit: the driver gives a bench run
# 600 s, so the launcher gives up (and NAMES the stuck ranks) well before that; DVO_PG_TIMEOUT (300 s) bounds the rendezvous itself.
DEFAULT_TIMEOUT_S = 420.0


def default_timeout() -> float:
    """Seconds after which `launch_ranks` gives up (DVO_LAUNCH_TIMEOUT overrides; <= 0 means no limit)."""
    try:
        return float(os.environ.get("DVO_LAUNCH_TIMEOUT", DEFAULT_TIMEOUT_S))
    except ValueError:
        return DEFAULT_TIMEOUT_S


def die_with_parent(sig: Optional[int] = None) -> bool:
    """Linux: deliver `sig` (default SIGKILL) to THIS process when its parent dies (prctl PR_SET_PDEATHSIG), so that a launcher
    killed with SIGKILL - which can run no handler - still takes its children with it.  Returns False where prctl is missing."""
    import ctypes
    import signal
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        return libc.prctl(1, int(signal.SIGKILL if sig is None else sig), 0, 0, 0) == 0        # 1 = PR_SET_PDEATHSIG
    except (OSError, AttributeError):
        return False


def mark_rank(state: str, env=None) -> None:
    """Inside a rank: leave a `rank<k>.<state>` marker ("started", "done") in the launcher's status directory, so that a
    launcher that times out can name the ranks that never started or never finished.  No-op without a launcher.
    "started" also ties the rank's life to its parent's (the elastic agent): see die_with_parent."""
    env = os.environ if env is None else env
    d = env.get("DVO_LAUNCH_STATUS_DIR")
    if not d:
        return
    if state == "started":
        die_with_parent()
    try:
        with open(os.path.join(d, "rank{}.{}".format(env.get("RANK", "0"), state)), "w") as f:
            f.write(str(os.getpid()))
    except OSError:
        pass


def phase(what: str, env=None) -> None:
    """Inside a rank: one line `rank k: <what>` on stderr (and the latest phase in the launcher's status directory), so that the
    tail of a killed multi-GPU run says where each rank stopped (aligner built / process group formed / timed loop done / ...)."""
    env = os.environ if env is None else env
    rank = env.get("RANK", "0")
    print("rank {}: {}".format(rank, what), file=sys.stderr, flush=True)
    d = env.get("DVO_LAUNCH_STATUS_DIR")
    if d:
        try:
            with open(os.path.join(d, "rank{}.phase".format(rank)), "w") as f:
                f.write(what)
        except OSError:
            pass


def _unfinished(status_dir: str, n_ranks: int) -> str:
    names = set(os.listdir(status_dir)) if os.path.isdir(status_dir) else set()
    never = [k for k in range(n_ranks) if "rank{}.started".format(k) not in names]
    stuck = [k for k in range(n_ranks) if "rank{}.started".format(k) in names and "rank{}.done".format(k) not in names]
    last = []
    for k in stuck:
        try:
            with open(os.path.join(status_dir, "rank{}.phase".format(k))) as f:
                last.append("rank {} after '{}'".format(k, f.read().strip()))
        except OSError:
            pass
    return "ranks that never reported in: {}; ranks that started but did not finish: {}{}".format(
        never or "none", stuck or "none", ("; last phase: " + ", ".join(last)) if last else "")


def _kill_group(proc: subprocess.Popen, grace: float = 5.0) -> None:
    """SIGTERM, then SIGKILL, to the whole process group of the launcher child (torch.distributed.run AND its ranks: killing
    only the parent would orphan the ranks with the GPUs still held)."""
    import signal
    for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 10.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            proc.wait(timeout=wait)
        except subprocess.TimeoutExpired:
            continue
        # the group leader is gone; make sure no rank of the group outlives it
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        return


def launch_ranks(n_ranks: int, script: str, argv: Sequence[str], need_devices: bool = True,
                 extra_env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Runs `script argv...` as n_ranks processes (one per GPU) under torch.distributed.run and returns the exit code.
    The children's stdout / stderr are inherited, so rank 0's JSON line is this process's output too.
    need_devices: fail (LaunchError) when fewer than n_ranks HIP devices are visible instead of degrading.
    timeout: seconds (default `default_timeout()`, 7 min: below the driver's own 10-minute limit; <= 0: none).  On expiry the whole process group of the launcher is
    terminated (no rank survives) and LaunchError names the ranks that did not report in / did not finish."""
    if n_ranks < 1:
        raise LaunchError("n_ranks must be >= 1")
    if need_devices:
        have = visible_devices()
        if have < n_ranks:
            raise LaunchError("{} GPUs requested but only {} HIP device(s) visible".format(n_ranks, have))
    if timeout is None:
        timeout = default_timeout()
    if timeout is not None and timeout <= 0:
        timeout = None
    import tempfile
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    if extra_env:
        env.update(extra_env)
    with tempfile.TemporaryDirectory(prefix="dvo_launch_") as status_dir:
        env["DVO_LAUNCH_STATUS_DIR"] = status_dir
        # --standalone: the rendezvous store picks its own port on 127.0.0.1 (no free-port probe that another process could win)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(n_ranks), script] + list(argv)
        # Own session / process group, so that _kill_group reaches the elastic agent AND its ranks.  The price: a signal sent to
        # THIS process's group (GNU timeout, a closed terminal, a harness' killpg) no longer reaches them by itself, so
        #   * SIGTERM / SIGHUP / SIGINT to this process are turned into an exception: the `except BaseException` below kills the group;
        #   * the agent is told to die with this process (PR_SET_PDEATHSIG, SIGTERM: it then stops its workers), and every rank ties
        #     itself to the agent the same way (mark_rank("started") -> die_with_parent) - that covers a SIGKILL of this process.
        import signal

        class _Signalled(BaseException):
            pass

        def _on_signal(signum, _frame):
            raise _Signalled(signum)

        def _child_setup():
            die_with_parent(signal.SIGTERM)

        old = {}
        for sg in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            try:
                old[sg] = signal.signal(sg, _on_signal)
            except (ValueError, OSError):                     # not the main thread: the pdeathsig chain still holds
                pass
        proc = subprocess.Popen(cmd, env=env, start_new_session=True, preexec_fn=_child_setup)
        try:
            return proc.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            why = _unfinished(status_dir, n_ranks)
            _kill_group(proc)
            raise LaunchError("{} rank(s) did not finish within {:.0f} s and were terminated ({})".format(n_ranks, timeout, why))
        except _Signalled as sgn:
            _kill_group(proc)
            raise SystemExit(128 + int(sgn.args[0]))
        except BaseException:
            _kill_group(proc)
            raise
        finally:
            for sg, h in old.items():
                try:
                    signal.signal(sg, h)
                except (ValueError, OSError):
                    pass


def pg_timeout():
    """timedelta for torch.distributed.init_process_group: a rank that cannot reach its peers fails with a message instead of
    waiting for the default 10-30 minutes."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("DVO_PG_TIMEOUT", "300")))
