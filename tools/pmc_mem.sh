#!/bin/bash
# Memory-pipeline PMC passes over the reduction kernels of the default bench (each pass its own run, counters only):
#   tools/pmc_mem.sh <tag> [lib.so]      -> gpurun_out/pmc_mem_<tag>.txt   (per-dispatch averages of the level-0 launches)
set -o pipefail
export TMPDIR=/tmp
TAG=$1; LIB=$2
OUT=gpurun_out/pmc_mem_$TAG
rm -rf $OUT; mkdir -p $OUT
[ -n "$LIB" ] && export DVO_HIP_LIB=$PWD/$LIB
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
pass() {   # name counters...
  local name=$1; shift
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $name "; then return; fi
  timeout -k 10 150 rocprofv3 --kernel-include-regex "k_reduce" --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.txt
  echo "pass $name done"
}
pass sq   SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pass sq2  SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU
# at most 2 counters of a TA / TD instance and 4 of a TCP / TCC instance fit one pass ("exceeds the capabilities of the hardware")
pass ta1  TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2  TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass ta3  TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
pass tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcp3 TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum
pass tcp4 TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum
pass utc1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
pass utc2 TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum
pass td   TD_TD_BUSY_sum TD_TC_STALL_sum
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
python3 - $OUT <<'PY' > gpurun_out/pmc_mem_$TAG.txt
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
wc -l gpurun_out/pmc_mem_$TAG.txt
