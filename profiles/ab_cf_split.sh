# (record of a dropped experiment: k2_cf_loop / k2_cf_finish and their FHX_CF_* switches are not in the tree any more; the result is
# profiles/history/r04_v_cf_split_ab.txt, the kernels are described where they were, in fhx_k2.hip)
# the converging classes as loop kernel + finish kernel against the fused kernel: parity files, then the C3 kernel summary of each
# variant on one box (FHX_CF_SPLIT=0 fused; loop / finish compiled for 5 or 7 waves per SIMD), digest of all p and q in every line
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/r04/v_tests.txt 2>&1; tail -3 gpurun_out/r04/v_tests.txt
run() { # tag env...
  T=$1; shift
  env "$@" FHX_BENCH_HASH=1 bash profiles/run_profile.sh r04/v_$T --steps 10 --warmup 3 --no-cpu-baseline --no-k3-stress > /dev/null 2>&1
  echo "== $T ($*)"; grep "k2_cf_\|k2_queue_by_count" gpurun_out/r04/v_${T}_kernel_stats.txt
  python -c "import json; d=json.load(open('gpurun_out/r04/v_${T}_bench.json')); print(d['ms_per_step'], d['kernels_ms']['k2_pvalue'], d.get('result_digest') or d.get('digest') or [k for k in d if 'hash' in k or 'digest' in k])"
}
run fused FHX_CF_SPLIT=0
run s77 FHX_CF_LOOP_WAVES=7 FHX_CF_FIN_WAVES=7
run s57 FHX_CF_LOOP_WAVES=5 FHX_CF_FIN_WAVES=7
run s75 FHX_CF_LOOP_WAVES=7 FHX_CF_FIN_WAVES=5
run s55 FHX_CF_LOOP_WAVES=5 FHX_CF_FIN_WAVES=5
