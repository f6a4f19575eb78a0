#!/usr/bin/env python3
"""Summarises tools/step_valu_budget.sh: per kernel of the bench step, wave-instructions (VALU / SALU / LDS / VMEM), waves, busy
cycles, HBM bytes, and its share of the whole command; plus durations from the kernel trace.

    step_valu_budget.py <dir> <steps> <warmup>

The command `bench.py --steps K --warmup W --no-extras` runs W + K pipelined steps, min(K,10) step-synchronous steps and
min(K,5) roofline-leg solves of ONE of the four slices: W + K + min(K,10) + min(K,5)/4 step-equivalents; `per step` divides by that."""
import csv
import glob
import re
import sys
from collections import defaultdict

root, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
step_equiv = warm + steps + min(steps, 10) + min(steps, 5) / 4.0


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name


def grid_of(r):
    return r.get("Grid_Size") or r.get("Grid_Size_X") or "?"


counters = defaultdict(lambda: defaultdict(float))     # kernel -> counter -> total
calls = defaultdict(int)
for p in sorted(glob.glob(root + "/*/")):
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fp:
            rows = list(csv.DictReader(fp))
        seen = defaultdict(lambda: defaultdict(int))
        for r in rows:
            k = short(r["Kernel_Name"])
            if k.startswith("k_reduce"):                 # one instantiation serves every level: tell the launches apart by their grid
                k += " grid=" + grid_of(r)
            counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k][r["Counter_Name"]] += 1
        for k, s in seen.items():
            calls[k] = max(calls[k], max(s.values()))

dur = {}
trace_files = glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True)
if trace_files:                                          # per-dispatch trace: durations by (kernel, grid)
    acc = defaultdict(lambda: [0, 0.0])
    for f in trace_files:
        with open(f) as fp:
            for r in csv.DictReader(fp):
                n = short(r.get("Kernel_Name", ""))
                if not n.startswith("k_"):
                    continue
                if n.startswith("k_reduce"):
                    n += " grid=" + grid_of(r)
                acc[n][0] += 1
                acc[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for n, (c, t) in acc.items():
        dur[n] = (c, t, t / max(c, 1))
else:
    for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
        with open(f) as fp:
            for r in csv.DictReader(fp):
                n = short(r.get("Name", ""))
                if n.startswith("k_"):
                    dur[n] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3)

print(open(root + "/command.txt").read().strip())
print("step-equivalents in this command: %.2f (warm-up %d + timed %d + step-synchronous %d + roofline leg %d/4)" %
      (step_equiv, warm, steps, min(steps, 10), min(steps, 5)))
tot_valu = sum(c.get("SQ_INSTS_VALU", 0.0) for c in counters.values())
tot_busy = sum(c.get("SQ_BUSY_CYCLES", 0.0) for c in counters.values())
tot_dur = sum(v[1] for v in dur.values())
print("total VALU wave-instructions: %.4g (%.4g per step); total kernel time %.1f ms (%.3f ms per step, summed over streams)" %
      (tot_valu, tot_valu / step_equiv, tot_dur / 1e3, tot_dur / 1e3 / step_equiv))
print()
hdr = "%-64s %7s %11s %7s %10s %10s %10s %10s %9s %9s %10s %8s %7s" % (
    "kernel", "calls", "VALU/step", "VALU%", "SALU/step", "LDS/step", "VMEMrd/st", "VMEMwr/st", "waves/st", "MB/step", "avg_us", "time%", "VALU/us")
print(hdr)
for k in sorted(counters, key=lambda k: -counters[k].get("SQ_INSTS_VALU", 0.0)):
    c = counters[k]
    hbm = 2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)        # KB (gfx950: FETCH_SIZE x2, MI355X_MICROARCH.md)
    d = dur.get(k, (0, 0.0, 0.0))
    print("%-64s %7d %11.4g %6.1f%% %10.4g %10.4g %10.4g %10.4g %9.4g %9.1f %10.1f %7.1f%% %7.0f" % (
        k[:64], calls[k], c.get("SQ_INSTS_VALU", 0) / step_equiv, 100 * c.get("SQ_INSTS_VALU", 0) / max(tot_valu, 1),
        c.get("SQ_INSTS_SALU", 0) / step_equiv, c.get("SQ_INSTS_LDS", 0) / step_equiv, c.get("SQ_INSTS_VMEM_RD", 0) / step_equiv,
        c.get("SQ_INSTS_VMEM_WR", 0) / step_equiv, c.get("SQ_WAVES", 0) / step_equiv, hbm * 1024 / 1e6 / step_equiv,
        d[2], 100 * d[1] / max(tot_dur, 1e-9), (c.get("SQ_INSTS_VALU", 0) / max(d[1], 1e-9))))
print()
print("wave states (share of SQ_WAVE_CYCLES-equivalent: SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY are wave-cycle sums)")
for k in sorted(counters, key=lambda k: -counters[k].get("SQ_INSTS_VALU", 0.0)):
    c = counters[k]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    print("  %-56s active_any %.2f  active_valu %.2f  wait_inst %.2f  wait_any %.2f  lds_conflict_cycles/step %.3g" % (
        k[:56], c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
        c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_LDS_BANK_CONFLICT", 0) / step_equiv))
print()
print("per launch (averages over the dispatches of the counter passes; HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction):")
for k in sorted(counters, key=lambda k: -counters[k].get("SQ_INSTS_VALU", 0.0)):
    if not k.startswith(("k_reduce", "k_gn_small")):
        continue
    c, n = counters[k], max(calls[k], 1)
    print("  %-64s VALU %.4g  VMEM_RD %.4g  waves %.4g  FETCH_SIZE_KB %.6g  WRITE_SIZE_KB %.6g  HBM bytes %.6g  avg_us %.2f" % (
        k[:64], c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_VMEM_RD", 0) / n, c.get("SQ_WAVES", 0) / n, c.get("FETCH_SIZE", 0) / n,
        c.get("WRITE_SIZE", 0) / n, (2.0 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / n, dur.get(k, (0, 0, 0))[2]))
