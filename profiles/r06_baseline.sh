# round-6 starting point: timeline of a 1/8 shard and of the full pass, host call times, C2 pass
mkdir -p gpurun_out/r06
bash profiles/shard_timeline.sh 8 > gpurun_out/r06/a_tl_shard8.txt 2>&1
bash profiles/shard_timeline.sh 1 > gpurun_out/r06/a_tl_full.txt 2>&1
FHX_CALL_TIMES=1 FHX_FIT_TIMES=1 python bench.py --shard-of 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --no-k3-stress 2>&1 >/dev/null | grep -v "^\[" | tail -12 > gpurun_out/r06/a_call_times_shard8.txt
FHX_CALL_TIMES=1 FHX_FIT_TIMES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --no-k3-stress 2>&1 >/dev/null | grep -v "^\[" | tail -12 > gpurun_out/r06/a_call_times_c3.txt
python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress > gpurun_out/r06/a_c2_bench.json 2>/dev/null
python bench.py --shard-of 8 --steps 40 --warmup 5 --no-cpu-baseline --no-parity-check --no-k3-stress > gpurun_out/r06/a_shard8_bench.json 2>/dev/null
grep -v "^W2026\|^E2026" gpurun_out/r06/a_tl_shard8.txt | head -50
cat gpurun_out/r06/a_call_times_shard8.txt gpurun_out/r06/a_call_times_c3.txt
python -c "
import json
for f in ('a_c2_bench','a_shard8_bench'):
    d=json.load(open('gpurun_out/r06/%s.json'%f)); print(f, d['ms_per_step'], d.get('ms_per_pass'), d['kernels_ms'])"
