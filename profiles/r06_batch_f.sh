#!/bin/bash
mkdir -p gpurun_out/r06
(timeout 900 python -m pytest tests/test_gpu_nonfixed.py -x -q 2>&1 | tail -15) > gpurun_out/r06/f_nonfixed_tests.txt
cat gpurun_out/r06/f_nonfixed_tests.txt
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "nonfixed or offgrid or f8 or f11" 2>&1 | tail -5) > gpurun_out/r06/f_nonfixed_goldens.txt
cat gpurun_out/r06/f_nonfixed_goldens.txt
python profiles/nonfixed_pairs_time.py > gpurun_out/r06/f_nonfixed_pairs_time.txt 2>&1; tail -12 gpurun_out/r06/f_nonfixed_pairs_time.txt
# what makes the sharded schedule (one rank over RCCL) 1.2 ms slower than the plain pass of the same shard?
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for v in "FHX_X=default" "FHX_NO_SPIN=1" "FHX_Q_PREFILL=0" "FHX_K3_DENSE=0"; do
  env $v FHX_FORCE_DIST=1 python bench.py --shard-of 8 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-weak --no-k3-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v forced dist ms/pass %.3f' % d['ms_per_step'], d['kernels_ms'], d.get('stage_ms'))"
done 2>&1 | tee gpurun_out/r06/f_forced_dist.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_fd && FHX_FORCE_DIST=1 rocprofv3 --kernel-trace -d /tmp/prof_fd -o run -- python $GRAFT_REPO_ROOT/bench.py --shard-of 8 --steps 4 --warmup 2 --no-cpu-baseline --no-parity-check --no-weak --no-k3-stress > /tmp/prof_fd.log 2>&1; cd $GRAFT_REPO_ROOT; python profiles/pass_timeline.py "$(find /tmp/prof_fd -name '*.db' | head -1)" > gpurun_out/r06/f_tl_forced_dist.txt 2>&1; grep -v "^W2026\|^E2026" gpurun_out/r06/f_tl_forced_dist.txt | head -70
