for i in 1 2 3; do
  for m in onepass exact; do
    if [ $m = exact ]; then export DVO_ST_SMALL_EXACT=1; else unset DVO_ST_SMALL_EXACT; fi
    timeout -k 10 300 python bench.py --no-extras --weighting student_t_reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['value']), round(d['ms_per_step'],3))" || exit 1
  done
done
