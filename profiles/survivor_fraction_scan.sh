#!/bin/bash
# Which synthetic rate structure puts which share of the rows below the exact BH cutoff (what K3 has to sort)?
IFS=$'\n'; for cfg in "--hotspots 0.2:4.9" "--hotspots 0.25:3.9" "--hotspots 0.3:3.3" "--hotspots 0.15:6.5" "--hotspots 0.2:4.9 --keep 0.9"; do unset IFS
  timeout 300 python bench.py $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check --no-k3-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg: pairs %d sorted %d frac %.3f  pass %.2f ms K3 %.3f ms' % (d['config']['pairs'], d['bh_rows_sorted_rank0'], d['bh_rows_sorted_rank0']/d['config']['pairs'], d['ms_per_step'], d['kernels_ms']['k3_bh_sort_scan']))"
done
