for cfg in "0.66 1.0" "0.9 1.0" "0.9 1.5" "0.97 1.0" "0.97 1.5" "0.8 0.7"; do
  set -- $cfg
  python bench.py --keep $1 --overdispersion $2 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check --no-k3-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('keep $1 s $2: pairs %d sorted %d frac %.3f  pass %.2f ms K3 %.3f ms' % (d['config']['pairs'], d['bh_rows_sorted_rank0'], d['bh_rows_sorted_rank0']/d['config']['pairs'], d['ms_per_step'], d['kernels_ms']['k3_bh_sort_scan']))"
done
