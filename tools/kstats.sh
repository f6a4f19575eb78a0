#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run, our kernels only: tools/kstats.sh [bench args]
export TMPDIR=/tmp
OUT=gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("k_", "void k_"))]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    print("%-58s calls %6s  total %9.3f ms  avg %9.1f us  %5.1f %%" % (r["Name"][:58], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
