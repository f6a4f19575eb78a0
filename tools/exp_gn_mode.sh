#!/bin/bash
# The true Gauss-Newton mode against the reference's V1: accuracy on the fixture excerpt and a rendered sequence, and throughput.
O=gpurun_out/gn_mode; rm -rf $O; mkdir -p $O
for cfg in "v1 10" "v1 3" "gn 10" "gn 5" "gn 3" "gn 2" "v3 10" "v3 3"; do
  set -- $cfg
  python tools/run_sequence.py --fixture --variant $1 --iterations $2 --out $O/fx_$1_$2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fixture   variant $1 iterations $2: ATE %.3f mm  RPE %.3f mm  GN iterations/pair %.1f' % (1e3 * d['ate_rmse_m'], 1e3 * d['rpe_trans_rmse_m'], d['gn_iterations_mean']))
" | tee -a $O/summary.txt
  python tools/run_sequence.py --synthetic 257 --variant $1 --iterations $2 --out $O/sy_$1_$2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('synthetic variant $1 iterations $2: ATE %.3f mm  RPE %.3f mm  GN iterations/pair %.1f' % (1e3 * d['ate_rmse_m'], 1e3 * d['rpe_trans_rmse_m'], d['gn_iterations_mean']))
" | tee -a $O/summary.txt
done
for cfg in "v1 10" "gn 10" "gn 3"; do
  set -- $cfg
  python bench.py --variant $1 --iterations $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench     variant $1 iterations $2: %.0f frame-pairs/s  %.2f ms/step  L0 %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))
" | tee -a $O/summary.txt
done
