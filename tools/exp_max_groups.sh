#!/bin/bash
# In-job sweep of the reduction's workgroup length (record groups per workgroup: DVO_MAX_GROUPS caps it, DVO_TARGET_BLOCKS sets the
# number of workgroups a launch aims for: experiments build only, make -C dense-visual-odometry_amd/csrc experiments; DVO_MAX_GROUPS is a
# product knob).
#   tools/exp_max_groups.sh "16 32 48" "1024 2048"
MGS=${1:-"16 32 48 64"}; TBS=${2:-"1024"}
for round in 1 2; do
for tb in $TBS; do for mg in $MGS; do
  DVO_HIP_LIB=$PWD/ab_libs/libdvo_hip_exp.so DVO_MAX_GROUPS=$mg DVO_TARGET_BLOCKS=$tb timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('max_groups $mg target_blocks $tb round $round: %.0f pairs/s  %.3f ms/step  L0 %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))" || exit 1
done; done; done
