#!/bin/bash
# k2h_heavy variants (rows per lane x waves per SIMD; "legacy" = round 1's per-lane kernel): heavy launch time from bench.py's
# HIP events on C3-synth, and a digest of every p and q of the run (all variants must print the same one).
#   gpurun -- 'bash profiles/heavy_variants.sh > gpurun_out/r03/heavy_variants.txt'
for v in "1 8" "2 8" "4 4" "4 3" "legacy 0"; do
  set -- $v
  if [ "$1" = legacy ]; then export FHX_K2_LEGACY=1; else unset FHX_K2_LEGACY; fi
  FHX_BENCH_HASH=1 FHX_K2H_ROWS=$1 FHX_K2H_WAVES=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rows/lane $1 waves/SIMD $2: heavy %.3f ms (fp64 issue frac %.3f), K2 %.3f ms, pass %.3f ms, digest of all p and q %s' % (1e3*d['roofline']['launch_seconds'], d['roofline']['fp64_valu_issue_frac'], d['kernels_ms']['k2_pvalue'], d['ms_per_step'], d.get('result_digest')))"
done
