#!/bin/bash
# A/B of kernel variants selected by environment variables, same box, same rows: per variant the pass time, the three kernel
# groups, the heavy launch and a digest of every p and q (must be equal across variants).
#   bash profiles/ab_env.sh "<bench args>" "VAR=a" "VAR=b OTHER=c" ...
ARGS=$1; shift
for v in "$@"; do
  env $v FHX_BENCH_HASH=1 python bench.py $ARGS --no-cpu-baseline --no-parity-check --no-k3-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s pass %.3f ms  K1 %.3f K2 %.3f K3 %.3f  heavy %.3f ms  sorted %s  digest %s  sort %s' % ('$v', d['ms_per_step'], d['kernels_ms']['k1_classify_hist'], d['kernels_ms']['k2_pvalue'], d['kernels_ms']['k3_bh_sort_scan'], 1e3*d['roofline']['launch_seconds'], d.get('bh_rows_sorted_rank0'), d.get('result_digest'), d.get('bh_sort_rank0')))"
done
