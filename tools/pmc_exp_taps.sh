#!/bin/bash
# needs the experiment build of the library: make -C dense-visual-odometry_amd/csrc experiments
export DVO_HIP_LIB=${DVO_HIP_LIB:-$PWD/ab_libs/libdvo_hip_exp.so}
# PMC passes over the tap formulations of the level-0 reduction (tools/exp_taps.py: the product kernel and k_reduce_x<1..4> in one
# process, so every pass counts all of them under the same conditions).  Each pass is its own run, counters only.
#   tools/pmc_exp_taps.sh            -> gpurun_out/pmc_exp_taps.txt (per-dispatch averages of the level-0 launches of each kernel)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_exp_taps
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/exp_taps.py --levels 0 --rounds 1 --launches 3 ${EXP_ARGS}"
pass() {   # name counters...
  local name=$1; shift
  timeout -k 10 170 rocprofv3 --kernel-include-regex "k_reduce" --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.txt
  echo "pass $name done"
}
pass sq   SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass sq2  SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD
pass ta1  TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2  TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
pass td   TD_TD_BUSY_sum TD_TC_STALL_sum
pass tcc  TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
python3 - $OUT <<'PY' > gpurun_out/pmc_exp_taps.txt
import csv, glob, sys, re
from collections import defaultdict
root = sys.argv[1]
for p in sorted(glob.glob(root + "/*/")):
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(list))
        for r in rows:
            k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
            agg[(k, r.get("Grid_Size") or r.get("Grid_Size_X"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== pass", p.rstrip("/").split("/")[-1])
        for (k, grid), c in sorted(agg.items()):
            print("  %s grid=%s n=%d" % (k, grid, len(next(iter(c.values())))))
            for name, v in sorted(c.items()):
                print("      %-44s %.6g" % (name, sum(v) / len(v)))
PY
cat $OUT/errors.txt 2>/dev/null
find $OUT -name "*.csv" -size +256k -delete; find $OUT -name "*.db" -delete
wc -l gpurun_out/pmc_exp_taps.txt
