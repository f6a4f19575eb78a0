#!/bin/bash
# A/B of the reduction kernel alone (fixed poses, 256 VGA pairs, host-mode launches) for several builds of
# libdvo_hip.so inside ONE gpurun call: tools/ab_reduce.sh a.so b.so ...   (three interleaved rounds)
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib round $round: " | tee -a gpurun_out/ab_reduce.log
    DVO_EXP_LAUNCHES=40 DVO_HIP_LIB=$PWD/$lib python tools/exp_ablate.py 2>/dev/null | tee -a gpurun_out/ab_reduce.log
  done
done
