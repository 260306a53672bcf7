#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_native_dist.py tests/test_gpu_fuzz.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r04/l_tests.txt 2>&1
tail -3 gpurun_out/r04/l_tests.txt | cut -c1-300
bash profiles/ab_env.sh "--steps 40 --warmup 5 --shard-of 8" "FHX_K3_SMALL=0" "FHX_K3_SMALL=1" "FHX_K3_SMALL=0" "FHX_K3_SMALL=1" > gpurun_out/r04/l_small_ab.txt 2>&1
bash profiles/ab_env.sh "--steps 20 --warmup 5" "FHX_K3_SMALL=0" "FHX_K3_SMALL=1" >> gpurun_out/r04/l_small_ab.txt 2>&1
cat gpurun_out/r04/l_small_ab.txt
bash profiles/run_profile.sh r04/l_s8 --steps 20 --warmup 5 --shard-of 8 --no-cpu-baseline --no-parity-check > gpurun_out/r04/l_s8_profile.log 2>&1
head -30 gpurun_out/r04/l_s8_kernel_stats.txt
