#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel stats + PMC passes) for this repo's kernels."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "")[-64:]


print(open(root + "/command.txt").read().strip())
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== rocprofv3 --kernel-trace --stats (this repo's kernels; torch kernels of the synthetic-stream renderer omitted)")
    with open(f) as fp:
        rows = list(csv.DictReader(fp))
    print("  %-64s %8s %14s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
    for r in rows:
        n = r.get("Name", "")
        if n.startswith(("k_", "void k_")):
            print("  %-64s %8s %14.1f %12.2f %12.2f %12.2f" % (short(n), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                            float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))

for f in glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True):
    with open(f) as fp:
        rows = list(csv.DictReader(fp))
    agg = defaultdict(list)
    meta = {}
    for r in rows:
        n = r.get("Kernel_Name", "")
        if "k_reduce" in n:
            key = short(n) + " grid=" + str(r.get("Grid_Size_X") or r.get("Grid_Size"))       # one instantiation serves every level
            agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[key] = (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"),
                         r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    print("== k_reduce dispatch metadata (grid, wg, arch_vgpr, accum_vgpr, sgpr, lds, scratch) and median duration")
    for k, v in sorted(agg.items()):
        v = sorted(v)
        print("  %-64s %s  n=%d median_us=%.2f" % (k, meta[k], len(v), v[len(v) // 2] / 1e3))

for p in sorted(glob.glob(root + "/pmc_*")):
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fp:
            rows = list(csv.DictReader(fp))
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for r in rows:
            k = short(r.get("Kernel_Name", ""))
            if "k_reduce" in k:
                k += " grid=" + str(r.get("Grid_Size") or r.get("Grid_Size_X"))
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        print("== PMC pass %s (per-dispatch averages)" % p.split("/")[-1])
        for k, c in sorted(agg.items()):
            line = "  %-64s" % k
            for name, val in sorted(c.items()):
                line += "  %s=%.4g" % (name, val / cnt[(k, name)])
            print(line + "  (n=%d)" % max(cnt[(k, nm)] for nm in c))
