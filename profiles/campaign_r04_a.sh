#!/bin/bash
# round 4, first GPU call: the tests this round touched, the headline line, C5 / C3w WITH their parity checks (streamed p / q),
# and K3 under a heavy small-p tail (--overdispersion 1.0) with its rocprofv3 kernel summary
mkdir -p gpurun_out/r04
{ free -g | head -2; nproc; rocm-smi --showmeminfo vram | grep -i total | head -1; } > gpurun_out/r04/a_box.txt 2>&1
{
echo "== bench + scale GPU tests"; timeout 1500 python -m pytest tests/test_gpu_bench.py tests/test_gpu_scale.py -x -q -m gpu --durations=8 2>&1 | tail -16
} > gpurun_out/r04/a_tests.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/a_c3_bench.json 2> gpurun_out/r04/a_c3_bench.err
timeout 1200 python bench.py --config C5 --steps 3 --warmup 1 > gpurun_out/r04/a_c5_bench.json 2> gpurun_out/r04/a_c5_bench.err
timeout 900 python bench.py --config C3w --steps 5 --warmup 2 > gpurun_out/r04/a_c3w_bench.json 2> gpurun_out/r04/a_c3w_bench.err
bash profiles/run_profile.sh r04/a_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline > gpurun_out/r04/a_od1_profile.log 2>&1
cat gpurun_out/r04/a_box.txt gpurun_out/r04/a_tests.txt
for f in a_c3 a_c5 a_c3w a_od1; do python -c "
import json
d=json.loads(open('gpurun_out/r04/${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['kernels_ms'], d.get('bh_rows_sorted_rank0'), (d.get('parity_check') or {}))"; done
tail -30 gpurun_out/r04/a_od1_kernel_stats.txt
