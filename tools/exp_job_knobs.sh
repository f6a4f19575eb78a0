#!/bin/bash
# In-job sweeps of the product library's tuning knobs (default bench, two rounds): streams per GPU, which levels take the persistent
# per-pair kernel (DVO_SMALL_MAX_PIX), its workgroup size (DVO_SMALL_THREADS).
run() {  # label, env..., -- bench flags
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: %.0f pairs/s  %.3f ms/step  L0 %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))" || exit 1
}
for round in 1 2; do
  for s in 3 4 5 6 8; do run "streams $s round $round" X=1 -- --streams $s; done
  for px in 4800 19200 76800; do run "small_max_pix $px round $round" DVO_SMALL_MAX_PIX=$px -- ; done
  for t in 256 512; do run "small_threads $t round $round" DVO_SMALL_THREADS=$t -- ; done
done
