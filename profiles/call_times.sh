mkdir -p gpurun_out/r04
FHX_CALL_TIMES=1 FHX_FIT_TIMES=1 python bench.py --shard-of 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --no-k3-stress 2>&1 >/dev/null | grep -v "^\[" | tail -12 > gpurun_out/r04/s_call_times_shard8.txt
FHX_CALL_TIMES=1 FHX_FIT_TIMES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --no-k3-stress 2>&1 >/dev/null | grep -v "^\[" | tail -12 > gpurun_out/r04/s_call_times_c3.txt
cat gpurun_out/r04/s_call_times_shard8.txt gpurun_out/r04/s_call_times_c3.txt
