# A/B of the ticket wait (kernel stores into pinned memory + host spin) against hipStreamSynchronize, 1/8 shard of C3, C2 and C3
mkdir -p gpurun_out/r06
B="--steps 40 --warmup 5 --no-cpu-baseline --no-parity-check --no-k3-stress"
for rep in 1 2; do
for v in spin nospin; do
  if [ $v = nospin ]; then export FHX_NO_SPIN=1; else unset FHX_NO_SPIN; fi
  python bench.py --shard-of 8 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v shard8 ms', round(d['ms_per_step'],4), d['kernels_ms'])"
  python bench.py --config C2 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v C2 ms', round(d['ms_per_step'],4), d.get('ms_per_pass'))"
done
done
unset FHX_NO_SPIN
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-k3-stress 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 ms', round(d['ms_per_step'],4), d['kernels_ms'], d['parity_check']['ok'])"
bash profiles/shard_timeline.sh 8 2>&1 | grep -v "^W2026\|^E2026" | head -45
