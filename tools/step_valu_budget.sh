#!/bin/bash
# Whole-step instruction budget: PMC passes over EVERY kernel of the default bench step, each pass its own
# run (counters only), then a kernel-trace pass for the durations.
#   tools/step_valu_budget.sh <tag> [bench args]   -> gpurun_out/valu_budget_<tag>/summary.txt
set -o pipefail
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/valu_budget_$TAG
rm -rf $OUT; mkdir -p $OUT
STEPS=5; WARM=1
ARGS="bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras $@"
echo "command: python3 $ARGS" > $OUT/command.txt
pass() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-include-regex "^(void )?k_" --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/errors.txt
  echo "pass $name done"
}
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass fetch FETCH_SIZE TCC_HIT_sum
pass write WRITE_SIZE TCC_MISS_sum
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "trace failed" >> $OUT/errors.txt
python3 tools/step_valu_budget.py $OUT $STEPS $WARM > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
cat $OUT/errors.txt 2>/dev/null
find $OUT -name "*.csv" -size +256k -delete; find $OUT -name "*.db" -delete
exit 0
