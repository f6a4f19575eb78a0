#!/bin/bash
# A/B of the whole job for several builds of libdvo_hip.so inside ONE gpurun call (three interleaved rounds):
#   tools/ab_job.sh [bench.py flags, e.g. --weighting huber] -- a.so b.so ...
flags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do flags+=("$1"); shift; done
shift
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in "$@"; do
    DVO_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "${flags[@]}" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib round $round: %.0f pairs/s  %.3f ms/step  L0 %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))" | tee -a gpurun_out/ab_job.log || exit 1
  done
done
