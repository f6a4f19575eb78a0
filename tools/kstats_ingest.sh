#!/bin/bash
# per-launch-shape timing of the ingest kernels (rocprofv3 kernel trace, grouped by kernel name and grid size)
export TMPDIR=/tmp
OUT=gpurun_out/kstats_ingest
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --kernel-include-regex "k_batch|k_list|k_ingest" --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"][:50], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for k, (n, t) in acc.items():
    print("%-50s grid %6s %5s %4s  calls %4d  avg %8.1f us" % (k[0], k[1], k[2], k[3], n, t / n))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
