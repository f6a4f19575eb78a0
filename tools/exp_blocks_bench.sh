#!/bin/bash
# needs the experiment build of the library: make -C dense-visual-odometry_amd/csrc experiments
export DVO_HIP_LIB=${DVO_HIP_LIB:-$PWD/ab_libs/libdvo_hip_exp.so}
# Workgroups per launch of the reduction (DVO_TARGET_BLOCKS / DVO_MAX_GROUPS), measured on the whole default job (4 streams).
for cfg in "default default" "1024 20" "1024 38" "1024 75" "1024 150" "512 75"; do
  set -- $cfg
  if [ "$1" = default ]; then unset DVO_TARGET_BLOCKS DVO_MAX_GROUPS; else export DVO_TARGET_BLOCKS=$1 DVO_MAX_GROUPS=$2; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('target $1 maxg $2: %.0f pairs/s  L0 %.1f us  levels %s' % (d['value'], r['avg_launch_us'], ' '.join(('%.1f' % v['avg_us']) if v['avg_us'] else '-' for v in r['per_level'].values())))
" | tee -a gpurun_out/exp_blocks.log
done
