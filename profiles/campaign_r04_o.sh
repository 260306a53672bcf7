#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/o_gpu_tests.txt 2>&1; tail -2 gpurun_out/r04/o_gpu_tests.txt
bash profiles/run_profile.sh r04/o_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline > gpurun_out/r04/o_od1_profile.log 2>&1
head -20 gpurun_out/r04/o_od1_kernel_stats.txt
bash profiles/run_pmc.sh r04/o_od1 --steps 3 --warmup 1 --overdispersion 1.0 --no-parity-check > gpurun_out/r04/o_od1_pmc.log 2>&1
grep -E "rs_|bh_|k3_|kernel " gpurun_out/r04/o_od1_pmc.txt
timeout 1200 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04/o_c5_bench.json 2> gpurun_out/r04/o_c5_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r04/o_c5_bench.json').read().strip().splitlines()[-1]); print('c5', d['value'], d['ms_per_step'], d['kernels_ms'], d['bh_rows_sorted_rank0'], d['parity_check']['ok'])"
