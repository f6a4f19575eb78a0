#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise.
for q in default 8 16 2; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for st in 4 8; do
  python bench.py --steps 10 --warmup 2 --streams $st --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('GPU_MAX_HW_QUEUES=$q streams=$st: %.0f pairs/s  %.2f ms/step  L0 %.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us']))
" | tee -a gpurun_out/exp_hwq.log
  done
done
# second sweep: stream counts (and batch sizes they divide) against the queue count
for cfg in "default 4 1024" "default 2 1024" "default 3 1023" "8 3 1023" "8 2 1024" "default 6 1020" "default 5 1020"; do
  set -- $cfg
  if [ "$1" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$1; fi
  python bench.py --steps 10 --warmup 2 --streams $2 --batch $3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('GPU_MAX_HW_QUEUES=$1 streams=$2 batch=$3: %.0f pairs/s  %.2f ms/step  L0 %.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us']))
" | tee -a gpurun_out/exp_hwq.log
done
