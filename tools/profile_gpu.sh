#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then PMC passes
# (each in its own run, as MI355X_MICROARCH.md prescribes).  Small summaries land in gpurun_out/prof/.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $@"
echo "command: python3 $ARGS" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-include-regex "k_reduce" --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-include-regex "k_reduce" --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-include-regex "k_reduce" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- python3 $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-include-regex "k_reduce" --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
python3 tools/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small artefacts (the raw per-dispatch CSVs are tens of MB)
find $OUT -name "*.csv" -size +512k -delete
find $OUT -name "*.db" -delete
