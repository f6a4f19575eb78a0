#!/bin/bash
# Does a working set that fits the Infinity Cache (256 MiB) make the level-0 reduction faster?  Same kernel, batch sizes whose
# per-iteration data (~2.95 MB per VGA pair) do / do not fit.
for cfg in "1024 4" "64 1" "32 1" "128 2" "64 2" "256 1" "128 1"; do
  set -- $cfg
  python bench.py --batch $1 --streams $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
ppl = $1 // $2
print('batch $1 streams $2: %.0f pairs/s  L0 %.1f us per launch of %d pairs = %.3f us/pair  levels %s' % (d['value'], r['avg_launch_us'], ppl, r['avg_launch_us'] / ppl, ' '.join(('%.1f' % v['avg_us']) if v['avg_us'] else '-' for v in r['per_level'].values())))
" | tee -a gpurun_out/exp_mall.log
done
