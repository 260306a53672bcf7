#!/bin/bash
mkdir -p gpurun_out/r04
bash profiles/ab_env.sh "--steps 20 --warmup 5" "FHX_CF_PERLANE=1" "FHX_CFU_ROWS=4" "FHX_CFU_ROWS=2" "FHX_CF_PERLANE=1" "FHX_CFU_ROWS=4" > gpurun_out/r04/h_cfu_ab.txt 2>&1
bash profiles/ab_env.sh "--steps 10 --warmup 3 --overdispersion 1.0" "FHX_CF_PERLANE=1" "FHX_CFU_ROWS=4" >> gpurun_out/r04/h_cfu_ab.txt 2>&1
bash profiles/ab_env.sh "--steps 5 --warmup 2 --config C5 --max-chroms 4" "FHX_CF_PERLANE=1" "FHX_CFU_ROWS=4" >> gpurun_out/r04/h_cfu_ab.txt 2>&1
cat gpurun_out/r04/h_cfu_ab.txt
bash profiles/run_profile.sh r04/h_c3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/h_c3_profile.log 2>&1
head -26 gpurun_out/r04/h_c3_kernel_stats.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -3
