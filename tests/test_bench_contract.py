"""bench.py prints ONE JSON line with the keys the driver and the judge read (GPU box only)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16",
                          "--cpu-pairs", "1"], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "frame-pairs/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    # the timed steps are pipelined windows; the step-synchronous time of the same steps is reported beside them and is not `value`
    ss = d["config"]["step_synchronous"]
    assert "pipelined" in d["config"]["stepping"] and ss["ms_per_step"] > 0 and ss["frame_pairs_per_s"] > 0
    assert abs(d["value"] - 16 / (1e-3 * d["ms_per_step"])) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["accuracy"]["ate_rmse_m"] - d["accuracy"]["reference_ate_rmse_m"]) < 1e-6
    # nothing in the line is hard-coded prose about counters: recorded readings sit under keys that say so, with their commit
    assert "limiter_measured" not in r
    rl = r["recorded_limiter_not_measured_in_this_run"]
    assert rl is None or ("commit" in rl and "source" in rl and isinstance(rl["stale"], bool))
    assert "traffic_recorded_at" in r and isinstance(r["traffic_stale"], bool)
    # the job-level roofline figure (SURVEY 8d bytes per frame pair x value) ...
    ws = r["whole_step"]
    assert "error" not in ws, ws
    assert 3e7 < ws["algorithmic_bytes_per_pair"] < 6e7 and len(ws["valid_fraction_per_level"]) == 4
    assert abs(ws["frac"] - ws["achieved"] / ws["peak"]) < 1e-12 and 0 < ws["frac"] < 1
    assert abs(ws["achieved"] * 1e9 - ws["algorithmic_bytes_per_pair"] * d["value"]) < 1e-6 * ws["achieved"] * 1e9
    # ... and the bound the dominant kernel really runs at, checkable by the driver: instruction count of the kernel that was built
    # x the cost of a wave-instruction measured in this run, beside the measured launch
    vi = r["valu_issue"]
    assert "error" not in vi, vi
    for k in ("kernel_sha256", "isa_matches_committed", "instr_per_point", "instr_per_point_static", "ns_per_wave_instr", "predicted_us",
              "measured_us", "frac_of_issue_bound"):
        assert k in vi, k
    assert 100 < vi["instr_per_point_static"] < 200 and 0.5 < vi["ns_per_wave_instr"] < 5.0
    assert abs(vi["measured_us"] - r["avg_launch_us"]) < 1e-9
    assert vi["isa_matches_committed"] is True, "profiles/kernel_isa.json is not of the kernel that was built: run tools/isa_count.py --write"
    # the rest of north_star's report, on the same clock (N = 1, default workload): config 3, PCIe-inclusive, Student-t
    c3 = d["config3"]
    assert "error" not in c3, c3
    assert c3["frame_pairs_per_s"] > 0 and c3["batch_pairs"] in (128, 256, 512, 1024) and "1280x960" in c3["workload"] and "Huber" in c3["workload"]
    r3 = c3["roofline"]
    for k in ("avg_launch_us", "algorithmic_bytes_per_launch", "achieved", "peak", "frac"):
        assert k in r3, k
    assert abs(r3["frac"] - r3["achieved"] / r3["peak"]) < 1e-12 and 0 < r3["frac"] < 1
    pc = d["pcie_inclusive"]
    assert "error" not in pc, pc
    for k in ("gray_2_slices", "gray_4_slices", "colour_2_slices", "colour_4_slices"):
        assert pc[k]["frame_pairs_per_s"] > 0 and pc[k]["link_GBs"] > 0
    stt = d["student_t_reference"]
    assert "error" not in stt, stt
    assert stt["frame_pairs_per_s"] > 0 and stt["gn_iterations_per_pair"] > 0
    sq = d["sequence_png"]
    assert "error" not in sq, sq
    assert sq["best"]["frame_pairs_per_s"] > 0 and sq["from_memory_frame_pairs_per_s"] > 0 and sq["leg_seconds"] < 60
    pm = d["production_mode"]
    assert pm["frame_pairs_per_s"] > 0 and "accuracy_note" not in pm and "recorded_accuracy_not_measured_in_this_run" in pm


def test_bench_through_the_rank_launcher_with_one_rank():
    """`bench.py --gpus 1 --spawn`: the multi-GPU entry rehearsed on one GPU — launcher -> torch.distributed.run -> one rank with a
    process group and the RCCL pose gather inside the timed region.  The line names world size 1, reports the gather time and the
    per-rank figures a SCALE record needs, and its value stays within 15 % of the plain single-process run."""
    def run(extra):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "256",
                              "--no-cpu-baseline", "--no-extras"] + extra, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])
    plain = run([])
    spawned = run(["--spawn", "--config3-all-ranks"])
    c3 = spawned["config3"]                               # the N > 1 form of the 1280x960 leg: every rank runs it, rank 0 gathers
    assert c3["n_gpus"] == 1 and len(c3["frame_pairs_per_s_per_rank"]) == 1 and c3["frame_pairs_per_s"] > 0
    assert 0 < c3["roofline_frac_per_rank"][0] < 1 and c3["level0_avg_launch_us_per_rank"][0] > 0
    assert spawned["n_gpus"] == 1 and spawned["ranks_joined"] == 1 and "world size 1" in spawned["collective"]
    assert spawned["config"]["pose_gather_ms_per_step"] is not None and spawned["config"]["pose_gather_ms_per_step"] >= 0
    pr = spawned["per_rank"]
    assert len(pr["ms_per_step"]) == 1 and len(pr["level0_avg_launch_us"]) == 1 and pr["ms_per_step_min"] <= pr["ms_per_step_max"]
    assert plain["collective"].startswith("none") and "per_rank" not in plain
    assert abs(spawned["value"] - plain["value"]) < 0.15 * plain["value"], (spawned["value"], plain["value"])
