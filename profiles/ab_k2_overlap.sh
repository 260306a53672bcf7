# (record of a dropped experiment: FHX_K2_OVERLAP and the second stream are not in the tree; result: profiles/history/r04_ov_k2_overlap_ab.txt)
# the converging classes on a lowest-priority stream beside k2h_heavy (FHX_K2_OVERLAP=1): both eligible at the same event, the heavy
# launch takes every register file first, the class kernels' workgroups start as its waves run out of tasks.  C3 and the 1/8 shard,
# digest of all p and q in every line.
mkdir -p gpurun_out/r04
for O in 0 1 0 1; do
  FHX_K2_OVERLAP=$O FHX_BENCH_HASH=1 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-parity-check --no-k3-stress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('C3 overlap=$O', d['ms_per_step'], d['kernels_ms'], [d[k] for k in d if 'digest' in k or 'hash' in k])"
  FHX_K2_OVERLAP=$O FHX_BENCH_HASH=1 python bench.py --shard-of 8 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-k3-stress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('1/8 overlap=$O', d['ms_per_step'], d['kernels_ms'], [d[k] for k in d if 'digest' in k or 'hash' in k])"
done
FHX_K2_OVERLAP=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldens or oracle_every_row or three_passes" 2>&1 | tail -2
