#!/bin/bash
mkdir -p gpurun_out/r06
(timeout 900 python -m pytest tests/test_gpu_nonfixed.py tests/test_gpu_writer.py -x -q 2>&1 | tail -6) > gpurun_out/r06/g_tests.txt; cat gpurun_out/r06/g_tests.txt
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for v in "FHX_X=default"; do
  env $v FHX_FORCE_DIST=1 python bench.py --shard-of 8 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-weak --no-k3-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v forced dist ms/pass %.3f' % d['ms_per_step'], d['kernels_ms']); print(d.get('stage_ms'))"
done 2>&1 | tee gpurun_out/r06/g_forced_dist.txt
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
# the heavy launch's own clock and the A/B of the clock sampling (digest must be equal)
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('C3 ms %.3f heavy %.3f ms clock %s GHz floor %.3f ms pass/floor %.3f' % (d['ms_per_step'], 1e3*r['launch_seconds'], r.get('shader_clock_ghz'), r['valu_issue_floor_ms'], r['pass_over_floor']))" | tee gpurun_out/r06/g_clock.txt
python profiles/cli_c5.py --check-rows 0 --again-with FHX_EMIT_WRITERS=1 --again-with FHX_EMIT_WRITERS=4 --again-with FHX_EMIT_WRITERS=16 > gpurun_out/r06/g_cli_c5.txt 2>&1
grep -E "^fithic|format \+ deflate|copy-out|output file|rows into the engine|row arrays" gpurun_out/r06/g_cli_c5.txt
