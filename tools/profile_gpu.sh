#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py. Outputs under gpurun_out/prof/.
set -o pipefail
export TMPDIR=/tmp
B=${1:-256}
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 1 --batch $B --no-cpu-baseline"
rocprofv3 --kernel-include-regex "k_" --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-include-regex "k_" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-include-regex "k_" --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-include-regex "k_" --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-include-regex "k_" --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1
python3 tools/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small artefacts
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
