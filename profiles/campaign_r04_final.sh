#!/bin/bash
# end-of-round evidence on the final tree: bench line (with cpu_baseline, parity_check, k3_stress) + rocprofv3 kernel summary of the
# same command, HBM traffic per kernel (two PMC passes), SQ counters (VALU issue share), every GPU test, the other configurations
# WITH their parity checks, smoke()
mkdir -p gpurun_out/r04
bash profiles/run_profile.sh r04/z --steps 20 --warmup 5 > gpurun_out/r04/z_profile.log 2>&1
bash profiles/run_pmc.sh r04/z --steps 3 --warmup 1 --no-parity-check > gpurun_out/r04/z_pmc.log 2>&1
bash profiles/run_pmc_counters.sh r04/z SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE > gpurun_out/r04/z_counters.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/z_gpu_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/z_smoke.txt 2>&1
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r04/z_c2_bench.json 2> gpurun_out/r04/z_c2_bench.err
timeout 900 python bench.py --config C3w --steps 5 --warmup 2 > gpurun_out/r04/z_c3w_bench.json 2> gpurun_out/r04/z_c3w_bench.err
timeout 1200 python bench.py --config C5 --steps 3 --warmup 1 > gpurun_out/r04/z_c5_bench.json 2> gpurun_out/r04/z_c5_bench.err
FHX_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-weak > gpurun_out/r04/z_forced_dist_bench.json 2> gpurun_out/r04/z_forced_dist_bench.err
tail -3 gpurun_out/r04/z_gpu_tests.txt; cat gpurun_out/r04/z_smoke.txt | tail -1
head -28 gpurun_out/r04/z_kernel_stats.txt
for f in z z_c2 z_c3w z_c5 z_forced_dist; do python -c "
import json
d=json.loads(open('gpurun_out/r04/${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['kernels_ms'], (d.get('parity_check') or {}).get('ok'), d.get('strong_efficiency'))"; done
# second call of the round's end: the scaling model (largest shard of an N-way sharding alone on this GPU, plain and under
# FHX_FORCE_DIST=1) and K3 under a heavy small-p tail with its kernel summary and PMC traffic
python profiles/scaling_model.py --config C3 --steps 40 > gpurun_out/r04/z_scaling_model.txt 2>&1
bash profiles/run_profile.sh r04/z_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline > gpurun_out/r04/z_od1_profile.log 2>&1
bash profiles/run_pmc.sh r04/z_od1 --steps 3 --warmup 1 --overdispersion 1.0 --no-parity-check > gpurun_out/r04/z_od1_pmc.log 2>&1
