#!/bin/bash
# One GPU call that refreshes the numbers quoted in DESIGN.md: default bench, other workloads, latency, PCIe, sequences,
# single-GPU rehearsal of the multi-rank path.  Outputs under gpurun_out/refresh/.
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --width 1280 --height 960 --levels 5 --weighting huber --batch 128 --steps 10 --no-cpu-baseline --no-extras > $O/bench_1280_huber.json 2>> $O/err.log
python bench.py --weighting student_t_reference --batch 256 --steps 10 --no-cpu-baseline --no-extras > $O/bench_student_t.json 2>> $O/err.log
python bench.py --weighting huber --steps 10 --no-cpu-baseline --no-extras > $O/bench_huber_vga.json 2>> $O/err.log
python bench.py --weighting student_t --batch 256 --steps 10 --no-cpu-baseline --no-extras > $O/bench_student_t_mean.json 2>> $O/err.log
python bench.py --variant v2 --steps 10 --no-cpu-baseline --no-extras > $O/bench_v2.json 2>> $O/err.log
python bench.py --variant v3 --steps 10 --no-cpu-baseline --no-extras > $O/bench_v3.json 2>> $O/err.log
python bench.py --width 1280 --height 960 --levels 5 --batch 128 --steps 10 --no-cpu-baseline --no-extras > $O/bench_1280_none.json 2>> $O/err.log
python bench.py --gpus 1 --spawn --steps 10 --no-cpu-baseline --no-extras 2>> $O/err.log | grep "^{" > $O/bench_dist1.json
python tools/latency_step.py > $O/latency.txt 2>> $O/err.log
python tools/exp_pcie.py > $O/pcie.txt 2>> $O/err.log
python tools/run_sequence.py --fixture --out $O/seq_fixture > $O/seq_fixture.txt 2>> $O/err.log
python tools/run_sequence.py --synthetic 2000 --out $O/seq_synth > $O/seq_synth.txt 2>> $O/err.log
python tools/run_sequence.py --synthetic 2000 --colour --out $O/seq_synth_bgr > $O/seq_synth_bgr.txt 2>> $O/err.log
python tools/run_sequence.py --synthetic 2000 --gpus 1 --spawn --out $O/seq_synth_spawn 2>> $O/err.log | grep "^{" > $O/seq_synth_spawn.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-28s %9.0f pairs/s  %6.2f ms/step  L0 %.1f us  frac %.3f  %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], " ".join("%.1f" % v["avg_us"] for v in r["per_level"].values() if v["avg_us"])))
PY
done
tail -n 12 $O/latency.txt $O/pcie.txt; tail -n 2 $O/seq_fixture.txt $O/seq_synth.txt $O/seq_synth_bgr.txt $O/seq_synth_spawn.txt
