#!/bin/bash
# A/B of builds of libdvo_hip.so inside ONE gpurun call (boxes differ by up to 10 %): tools/ab_bench.sh a.so b.so ...
# Prints frame-pairs/s and the level-0 reduction time per launch for each library, two interleaved rounds.
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "$@"; do
    DVO_HIP_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$lib round $round: %.0f pairs/s  L0 %.1f us  levels %s' % (d['value'], r['avg_launch_us'], ' '.join(('%.1f' % v['avg_us']) if v['avg_us'] else '-' for v in r['per_level'].values())))
" | tee -a gpurun_out/ab_bench.log
  done
done
