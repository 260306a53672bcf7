# host time inside the four calls of fhx_run_pass, spin against hipStreamSynchronize (FHX_NO_SPIN), 1/8 shard of C3 and C2
B="--steps 12 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress"
for v in spin nospin; do
  if [ $v = nospin ]; then export FHX_NO_SPIN=1; else unset FHX_NO_SPIN; fi
  echo "== $v shard8"; FHX_PASS_TIMES=1 python bench.py --shard-of 8 $B 2>&1 >/dev/null | grep fhx_run_pass | tail -6
  echo "== $v C2"; FHX_PASS_TIMES=1 python bench.py --config C2 $B 2>&1 >/dev/null | grep fhx_run_pass | tail -6
done
