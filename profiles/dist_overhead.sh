#!/bin/bash
# plain single-GPU pass against the sharded schedule with one rank over real RCCL (FHX_FORCE_DIST=1), on a chromosome-1-sized
# shard and on C3; then the wall time of every C-ABI call of a pass on a 1/18 shard and on C3 (profiles/stage_times.py).
#   gpurun -- 'bash profiles/dist_overhead.sh <tag>'      -> gpurun_out/<tag>_dist_overhead.txt, gpurun_out/<tag>_stage_times.txt
TAG=${1:-x}
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for mc in 1 0; do
  python bench.py --max-chroms $mc --steps 40 --warmup 5 --no-cpu-baseline --no-parity-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('max-chroms $mc rows', d['config']['pairs'], 'plain ms/pass %.3f' % d['ms_per_step'], d['kernels_ms'])"
  FHX_FORCE_DIST=1 python bench.py --max-chroms $mc --steps 40 --warmup 5 --no-cpu-baseline --no-parity-check --no-weak 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   forced dist (1 rank, RCCL) ms/pass %.3f' % d['ms_per_step'], d['kernels_ms'], d['stage_ms'])"
done 2>&1 | tee gpurun_out/${TAG}_dist_overhead.txt
python profiles/stage_times.py --max-chroms 3 2>&1 | grep -v amdgpu | tee gpurun_out/${TAG}_stage_times.txt
python profiles/stage_times.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/${TAG}_stage_times.txt
