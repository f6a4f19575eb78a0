#!/usr/bin/env python3
"""Coverage of a multi-stream run by its own kernels, from a rocprofv3 --kernel-trace CSV: over the last `frac` of the trace, the
share of wall time with at least one / two / ... of this repo's kernels running, and the idle share.
    python tools/trace_union.py <kernel_trace.csv> [frac]
    python tools/trace_union.py <kernel_trace.csv> window <kernel-name-substring> <i> <j>   # from the i-th to the j-th start of that kernel"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "window" else 0.6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows
            if r["Kernel_Name"].startswith(("k_", "void k_")))
if len(sys.argv) > 2 and sys.argv[2] == "window":
    starts = [e[0] for e in ev if sys.argv[3] in e[2]]
    t_lo, t_hi = starts[int(sys.argv[4])], starts[int(sys.argv[5])]
    ev = [(max(s, t_lo), min(e, t_hi), n) for s, e, n in ev if e > t_lo and s < t_hi]
else:
    t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * (1 - frac)
    ev = [e for e in ev if e[0] >= t_lo]
pts = []
for s, e, _ in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
depth, last, hist = 0, pts[0][0], {}
for t, d in pts:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d
    last = t
span = pts[-1][0] - pts[0][0]
print("span %.3f ms, %d dispatches" % (span / 1e6, len(ev)))
for k in sorted(hist):
    print("  %d kernel(s) in flight: %6.2f %% of the span (%.3f ms)" % (k, 100.0 * hist[k] / span, hist[k] / 1e6))
gaps = sorted((b for a, b in hist.items() if a == 0))
# idle intervals by length
idle, depth, last = [], 0, pts[0][0]
for t, d in pts:
    if depth == 0 and t > last:
        idle.append(t - last)
    depth += d
    last = t
idle.sort()
if idle:
    print("  idle intervals: n=%d, median %.1f us, max %.1f us, total %.3f ms" % (len(idle), idle[len(idle) // 2] / 1e3, idle[-1] / 1e3, sum(idle) / 1e6))

# time each kernel spends as the ONLY one in flight
import re
from collections import defaultdict
alone = defaultdict(int)
total = defaultdict(int)
active = {}
evs = []
for i, (s_, e_, n) in enumerate(ev):
    evs.append((s_, 0, i))
    evs.append((e_, 1, i))
evs.sort()
last = evs[0][0]
for t, kind, i in evs:
    if len(active) == 1:
        alone[next(iter(active.values()))] += t - last
    for n in active.values():
        total[n] += t - last
    if kind == 0:
        active[i] = re.sub(r"\(.*$", "", ev[i][2]).replace("void ", "")
    else:
        active.pop(i, None)
    last = t
print("  kernel                                        in flight (ms)   alone (ms)")
for n, t in sorted(total.items(), key=lambda kv: -kv[1]):
    print("  %-44s %10.3f %12.3f" % (n[:44], t / 1e6, alone[n] / 1e6))
