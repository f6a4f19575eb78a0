"""`bench.py --gpus N` / `run_sequence.py --gpus N` start their own ranks (dvo_amd/launch.py).  CPU checks: the launcher
really starts N processes that form one process group (stub worker, gloo), a rank whose WORLD_SIZE differs from --gpus is
an error, and asking for more GPUs than are visible fails loudly instead of measuring one GPU."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

from dvo_amd import launch

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "dense-visual-odometry_amd"

_STUB = r'''
import json, os, sys
sys.path.insert(0, os.environ["DVO_PKG_DIR"])
import torch
import torch.distributed as dist
from dvo_amd import launch
assert launch.started_by_launcher()
rank, local_rank, world = launch.rank_env()
launch.check_world(int(sys.argv[sys.argv.index("--gpus") + 1]))
dist.init_process_group(backend="gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item()), "argv": sys.argv[1:]}))
dist.destroy_process_group()
'''


def _run_driver(code, timeout=300):
    env = dict(os.environ, DVO_PKG_DIR=str(PKG), OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [1, 2])
def test_launcher_starts_n_ranks_and_relays_rank0_json(tmp_path, n):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(_STUB)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dvo_amd import launch\n"
            "sys.exit(launch.launch_ranks(%d, %r, ['--gpus', '%d', '--steps', '3'], need_devices=False))\n"
            % (str(PKG), n, str(stub), n))
    out = _run_driver(code)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["sum"] == n * (n + 1) / 2
    assert d["argv"] == ["--gpus", str(n), "--steps", "3"]          # the arguments reach every rank unchanged


def test_rank_failure_is_the_launchers_exit_code(tmp_path):
    stub = tmp_path / "bad_worker.py"
    stub.write_text("import sys; sys.exit(3)\n")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dvo_amd import launch\n"
            "sys.exit(1 if launch.launch_ranks(2, %r, [], need_devices=False) != 0 else 0)\n" % (str(PKG), str(stub)))
    out = _run_driver(code)
    assert out.returncode == 1


def test_timeout_terminates_every_rank_and_names_the_stuck_ones(tmp_path):
    """A rank that hangs (stuck rendezvous / RCCL init on hardware) must not hang the command: launch_ranks gives up after its
    timeout, raises LaunchError naming the ranks that did not finish, and leaves no process of the group behind."""
    pidfile = tmp_path / "pids.txt"
    stub = tmp_path / "sleepy_worker.py"
    stub.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, os.environ['DVO_PKG_DIR'])\n"
        "from dvo_amd import launch\n"
        "open(%r, 'a').write('%%d\\n' %% os.getpid())\n"
        "if os.environ['RANK'] == '0':\n"
        "    launch.mark_rank('started'); launch.mark_rank('done'); sys.exit(0)\n"
        "launch.mark_rank('started')\n"
        "launch.phase('waiting for a peer')\n"
        "time.sleep(600)\n" % str(pidfile))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dvo_amd import launch\n"
            "try:\n"
            "    launch.launch_ranks(2, %r, [], need_devices=False, timeout=25)\n"
            "except launch.LaunchError as e:\n"
            "    print('LAUNCH_ERROR', e); sys.exit(7)\n"
            "sys.exit(0)\n" % (str(PKG), str(stub)))
    out = _run_driver(code, timeout=120)
    assert out.returncode == 7, out.stdout[-2000:] + out.stderr[-2000:]
    assert "LAUNCH_ERROR" in out.stdout and "did not finish: [1]" in out.stdout, out.stdout
    assert "rank 1 after 'waiting for a peer'" in out.stdout, out.stdout            # the last phase the stuck rank reported
    import time
    time.sleep(1.0)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    for pid in pids:                                    # no rank survives the launcher
        alive = True
        try:
            os.kill(pid, 0)
            # a zombie of an already-reaped parent may linger for a moment; check its state
            state = Path("/proc/%d/status" % pid).read_text().split("State:")[1].split()[0]
            alive = state not in ("Z", "X")
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, "rank process %d survived the timeout" % pid


def _alive(pid):
    try:
        os.kill(pid, 0)
        state = Path("/proc/%d/status" % pid).read_text().split("State:")[1].split()[0]
        return state not in ("Z", "X")
    except (ProcessLookupError, FileNotFoundError):
        return False


@pytest.mark.parametrize("sig", ["SIGTERM", "SIGKILL"])
def test_a_signalled_driver_takes_its_ranks_with_it(tmp_path, sig):
    """The ranks live in their own session (so that the launcher can kill the whole group), which means a signal to the DRIVER's
    group - GNU timeout, a harness' killpg, a closed terminal - does not reach them by itself.  SIGTERM: the driver's handler
    kills the group.  SIGKILL (no handler can run): the elastic agent dies with the driver (PR_SET_PDEATHSIG) and every rank with
    the agent.  Either way no rank survives with a GPU still held."""
    import signal
    import time
    pidfile = tmp_path / "pids.txt"
    stub = tmp_path / "sleepy_worker.py"
    stub.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, os.environ['DVO_PKG_DIR'])\n"
        "from dvo_amd import launch\n"
        "launch.mark_rank('started')\n"
        "open(%r, 'a').write('%%d\\n' %% os.getpid())\n"
        "time.sleep(600)\n" % str(pidfile))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dvo_amd import launch\n"
            "sys.exit(launch.launch_ranks(2, %r, [], need_devices=False, timeout=300))\n" % (str(PKG), str(stub)))
    env = dict(os.environ, DVO_PKG_DIR=str(PKG), OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    drv = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = time.time()
        while time.time() - t0 < 120 and (not pidfile.exists() or len(pidfile.read_text().split()) < 2):
            time.sleep(0.5)
        pids = [int(x) for x in pidfile.read_text().split()]
        assert len(pids) == 2 and all(_alive(p) for p in pids)
        drv.send_signal(getattr(signal, sig))
        rc = drv.wait(timeout=60)
        assert rc != 0
        t0 = time.time()
        while time.time() - t0 < 30 and any(_alive(p) for p in pids):
            time.sleep(0.5)
        assert not any(_alive(p) for p in pids), "a rank survived its driver's %s" % sig
    finally:
        if drv.poll() is None:
            drv.kill()
        for p in (pidfile.read_text().split() if pidfile.exists() else []):
            try:
                os.kill(int(p), signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass


def test_default_timeout_is_below_the_drivers_limit(monkeypatch):
    monkeypatch.delenv("DVO_LAUNCH_TIMEOUT", raising=False)
    assert 0 < launch.default_timeout() <= 450           # the driver gives a bench run 600 s: the launcher must speak first
    monkeypatch.setenv("DVO_LAUNCH_TIMEOUT", "42")
    assert launch.default_timeout() == 42.0
    assert launch.pg_timeout().total_seconds() > 0


def test_world_size_mismatch_is_an_error():
    assert launch.check_world(2, env={"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"}) == 2
    with pytest.raises(launch.LaunchError):
        launch.check_world(8, env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert launch.rank_env(env={}) == (0, 0, 1)
    assert not launch.started_by_launcher(env={})
    assert launch.started_by_launcher(env={"RANK": "0", "WORLD_SIZE": "4"})


@pytest.mark.parametrize("script", ["bench.py", "tools/run_sequence.py"])
def test_more_gpus_than_devices_fails_loudly(script):
    """On a host with fewer than 64 devices `--gpus 64` must exit non-zero with a clear message and no JSON line."""
    import torch
    assert torch.cuda.device_count() < 64
    args = [sys.executable, str(ROOT / script), "--gpus", "64"]
    if script.endswith("run_sequence.py"):
        args.append("--fixture")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run(args, env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert out.returncode != 0
    assert "64 GPUs requested" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
