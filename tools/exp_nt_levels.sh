#!/bin/bash
# In-job check of non-temporal loads for the valid-point records, per pyramid level (DVO_NT_LEVELS bit mask; experiments build:
# make -C dense-visual-odometry_amd/csrc experiments).  profiles/r04_experiments.txt [13].
for round in 1 2; do for m in 0 1 3; do DVO_HIP_LIB=$PWD/ab_libs/libdvo_hip_exp.so DVO_NT_LEVELS=$m timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nt_levels $m round $round: %.0f pairs/s  %.3f ms/step  L0 %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))"; done; done
