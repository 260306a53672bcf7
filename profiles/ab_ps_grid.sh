mkdir -p gpurun_out/r04
FHX_DEBUG_GRID=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check --no-k3-stress 2>&1 >/dev/null | grep resident_grid > gpurun_out/r04/t_grids.txt
for G in 0 1024 2048 3584 7168; do
  FHX_PS_GRID=$G bash profiles/run_profile.sh r04/t_ps$G --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --no-k3-stress > /dev/null 2>&1
  echo "FHX_PS_GRID=$G" >> gpurun_out/r04/t_ps.txt; grep "k2_queue<1" gpurun_out/r04/t_ps${G}_kernel_stats.txt >> gpurun_out/r04/t_ps.txt
done
for R in 4 2; do
  FHX_K2H_ROWS=$R bash profiles/shard_timeline.sh 8 > gpurun_out/r04/t_tl_rows$R.txt 2>&1
  FHX_K2H_ROWS=$R python bench.py --shard-of 8 --steps 20 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('rows $R', d['ms_per_step'], d['kernels_ms'])" >> gpurun_out/r04/t_rows.txt
  grep "k2h_heavy\|pass span" gpurun_out/r04/t_tl_rows$R.txt >> gpurun_out/r04/t_rows.txt
done
cat gpurun_out/r04/t_grids.txt gpurun_out/r04/t_ps.txt gpurun_out/r04/t_rows.txt
