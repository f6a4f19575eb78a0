#!/bin/bash
# Iteration schedules (finest level first) of the GN mode: accuracy on a rendered sequence with exact ground truth, and throughput.
O=gpurun_out/gn_sched; rm -rf $O; mkdir -p $O
for s in "3,3,3,3" "2,2,2,2" "1,2,2,2" "1,1,2,2" "1,1,1,2" "1,1,1,1" "0,2,2,2" "0,1,2,2" "1,2,3,3"; do
  python tools/run_sequence.py --synthetic 257 --variant gn --level-iterations $s --out $O/sy_$s 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gn schedule %-9s rendered: ATE %.3f mm  RPE %.4f mm  iterations/pair %.1f' % ('$s', 1e3 * d['ate_rmse_m'], 1e3 * d['rpe_trans_rmse_m'], d['gn_iterations_mean']))
" | tee -a $O/summary.txt
  python tools/run_sequence.py --fixture --variant gn --level-iterations $s --out $O/fx_$s 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gn schedule %-9s fixture : ATE %.3f mm  RPE %.4f mm' % ('$s', 1e3 * d['ate_rmse_m'], 1e3 * d['rpe_trans_rmse_m']))
" | tee -a $O/summary.txt
  python bench.py --variant gn --level-iterations $s --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gn schedule %-9s bench   : %.0f frame-pairs/s  %.2f ms/step' % ('$s', d['value'], d['ms_per_step']))
" | tee -a $O/summary.txt
done
