#!/bin/bash
mkdir -p gpurun_out/r06
(timeout 900 python -m pytest tests/test_gpu_k3_sort.py tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_dist.py tests/test_gpu_native_dist.py -x -q -n 2 2>&1 | tail -6) > gpurun_out/r06/h_tests.txt; cat gpurun_out/r06/h_tests.txt
bash profiles/r06_ab_heavy_split.sh 2>&1 | tee gpurun_out/r06/h_ab_heavy_split.txt | cut -c1-150
for t in "h_c3 --steps 8 --warmup 2" "h_shard8 --shard-of 8 --steps 20 --warmup 3" "h_hot55 --hotspots 0.25:3.9 --steps 6 --warmup 2" "h_od1 --overdispersion 1.0 --steps 6 --warmup 2"; do
  set -- $t; tag=$1; shift
  bash profiles/run_profile.sh r06/$tag "$@" --no-cpu-baseline --no-parity-check --no-k3-stress > /dev/null 2>&1
  echo "== $tag"; grep -E "k3_compact|k3_fill|k2h_heavy|bh_apply|k1_prezero" gpurun_out/r06/${tag}_kernel_stats.txt | cut -c1-110
  python -c "
import json; d=json.load(open('gpurun_out/r06/${tag}_bench.json')); print('$tag ms %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernels_ms'].items()})"
done 2>&1 | tee gpurun_out/r06/h_compact_tiles.txt
bash profiles/shard_timeline.sh 8 > gpurun_out/r06/h_tl_shard8.txt 2>&1; grep -v "^W2026\|^E2026" gpurun_out/r06/h_tl_shard8.txt | head -34
