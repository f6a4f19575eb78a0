#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV of a single-stream run: per kernel name the number of dispatches and
total busy time, plus the idle time between consecutive dispatches (launch gaps), over the last `frac` of the trace.
    python tools/trace_gaps.py <kernel_trace.csv> [frac]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = [e for e in ev if e[2].startswith(("k_", "void k_"))]
ev = ev[int(len(ev) * (1 - frac)):]
busy = defaultdict(lambda: [0, 0])
gap_small = gap_big = 0
ngap = 0
for i, (s, e, n) in enumerate(ev):
    name = n.split("(")[0].replace("void ", "")
    busy[name][0] += 1
    busy[name][1] += e - s
    if i:
        g = s - ev[i - 1][1]
        if 0 < g < 50_000:
            gap_small += g
            ngap += 1
        elif g >= 50_000:
            gap_big += g
span = ev[-1][1] - ev[0][0]
print("span %.3f ms, %d dispatches" % (span / 1e6, len(ev)))
for k, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s n=%5d  busy %8.3f ms  avg %7.2f us" % (k, c, t / 1e6, t / 1e3 / c))
print("  gaps < 50 us between dispatches: n=%d total %.3f ms avg %.2f us; gaps >= 50 us (host): %.3f ms" % (ngap, gap_small / 1e6, gap_small / 1e3 / max(ngap, 1), gap_big / 1e6))
