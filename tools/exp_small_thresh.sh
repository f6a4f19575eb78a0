#!/bin/bash
# Which levels go to the persistent per-pair kernel at B = 1024 / 4 streams: threshold (pixels) x workgroup size.
for cfg in "default default" "76800 1024" "76800 512" "76800 256" "19200 512" "19200 1024" "default default"; do
  set -- $cfg
  if [ "$1" = default ]; then unset DVO_SMALL_MAX_PIX DVO_SMALL_THREADS; else export DVO_SMALL_MAX_PIX=$1 DVO_SMALL_THREADS=$2; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('max_pix $1 threads $2: %.0f pairs/s  L0 %.1f us  levels %s' % (d['value'], r['avg_launch_us'], ' '.join(('%.1f' % v['avg_us']) if v['avg_us'] else '-' for v in r['per_level'].values())))
" | tee -a gpurun_out/exp_small_thresh.log
done
