# same-box look at the class kernels after a change of how their work is handed out: parity tests, the C3 kernel summary, the
# timeline of a 1/8 shard, and the tile size of the converging classes forced to 1, 2, 4 quarters
mkdir -p gpurun_out/r04
T=${1:-q}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/r04/${T}_tests.txt 2>&1; tail -3 gpurun_out/r04/${T}_tests.txt
bash profiles/run_profile.sh r04/$T --steps 10 --warmup 3 --no-cpu-baseline --no-k3-stress > /dev/null 2>&1; head -16 gpurun_out/r04/${T}_kernel_stats.txt
bash profiles/shard_timeline.sh 8 > gpurun_out/r04/${T}_tl_shard8.txt 2>&1
python bench.py --shard-of 8 --steps 20 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress > gpurun_out/r04/${T}_shard8_bench.json 2>/dev/null
for Q in 1 2 4; do FHX_CF_QUARTERS=$Q bash profiles/shard_timeline.sh 8 > gpurun_out/r04/${T}_tl_shard8_q$Q.txt 2>&1; done
grep -h "k2_queue\|k2h_heavy\|pass span" gpurun_out/r04/${T}_tl_shard8*.txt | cut -c1-110
python -c "import json; d=json.load(open('gpurun_out/r04/${T}_shard8_bench.json')); print(d['ms_per_step'], d['kernels_ms'])"
