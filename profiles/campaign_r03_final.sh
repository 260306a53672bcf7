#!/bin/bash
# end-of-round evidence on the final tree: bench line (with cpu_baseline and parity_check) + rocprofv3 kernel summary of the same
# command, HBM traffic per kernel (two PMC passes), the other configurations, a fuzz campaign
mkdir -p gpurun_out/r03 gpurun_out
bash profiles/run_profile.sh r03/ac --steps 20 --warmup 5 > gpurun_out/r03/ac_profile.log 2>&1
bash profiles/run_pmc.sh r03/ac --steps 3 --warmup 1 --no-parity-check > gpurun_out/r03/ac_pmc.log 2>&1
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r03/ac_c2_bench.json 2> gpurun_out/r03/ac_c2_bench.err
timeout 600 python bench.py --config C3w --steps 5 --warmup 2 --no-cpu-baseline --no-parity-check > gpurun_out/r03/ac_c3w_bench.json 2> gpurun_out/r03/ac_c3w_bench.err
timeout 900 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check > gpurun_out/r03/ac_c5_bench.json 2> gpurun_out/r03/ac_c5_bench.err
{
echo "== test_gpu_fuzz 4000:4400"; FHX_FUZZ_SEEDS=4000:4400 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_native_dist 300:400"; FHX_FUZZ_SEEDS=300:400 timeout 900 python -m pytest tests/test_gpu_native_dist.py -x -q -m gpu 2>&1 | tail -2
} > gpurun_out/r03/ac_campaign.txt 2>&1
tail -5 gpurun_out/r03/ac_profile.log; cat gpurun_out/r03/ac_campaign.txt
for f in ac_c2 ac_c3w ac_c5; do python -c "
import json
d=json.loads(open('gpurun_out/r03/${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['kernels_ms'], (d.get('parity_check') or {}).get('ok'))"; done
