mkdir -p gpurun_out/r04
for G in 2048 1024 1536; do
  FHX_CL_SHARDS=$G FHX_BENCH_HASH=1 bash profiles/run_profile.sh r04/cl_$G --steps 8 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress > /dev/null 2>&1
  echo "== FHX_CL_SHARDS=$G"; grep "k2_classify\|k2_queue\|k2h_scatter\|k2h_heavy" gpurun_out/r04/cl_${G}_kernel_stats.txt
  python -c "import json; d=json.load(open('gpurun_out/r04/cl_${G}_bench.json')); print(d['ms_per_step'], d['kernels_ms']['k2_pvalue'], [d[k] for k in d if 'digest' in k or 'hash' in k])"
done
