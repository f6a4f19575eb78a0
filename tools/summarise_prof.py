#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel stats + PMC passes) per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    return name[-70:]


for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats:", f)
    with open(f) as fp:
        rows = list(csv.DictReader(fp))
    for r in rows[:14]:
        print("  %-72s calls %6s  total_ns %12s  avg_ns %10s  pct %6s" % (
            short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

# per-dispatch durations by grid size for the reduction kernel
for f in glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True):
    with open(f) as fp:
        rows = list(csv.DictReader(fp))
    agg = defaultdict(list)
    for r in rows:
        if "k_reduce" in r.get("Kernel_Name", ""):
            key = (short(r["Kernel_Name"]), r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"))
            agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== k_reduce dispatches by grid:")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("  %s grid=%s vgpr=%s sgpr=%s lds=%s  n=%d avg_us=%.1f" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v) / 1e3))

for p in sorted(glob.glob(root + "/pmc*")):
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fp:
            rows = list(csv.DictReader(fp))
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for r in rows:
            k = (short(r.get("Kernel_Name", "")), r.get("Grid_Size"))
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        print("== PMC:", f)
        for k, c in sorted(agg.items(), key=lambda kv: -max(kv[1].values()))[:8]:
            print("  %s grid=%s" % k)
            for name, val in c.items():
                n = cnt[(k, name)]
                print("      %-24s per-dispatch %.4g  (n=%d)" % (name, val / n, n))
