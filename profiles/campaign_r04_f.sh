#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_native_dist.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04/f_tests.txt 2>&1
tail -3 gpurun_out/r04/f_tests.txt | cut -c1-300
FHX_K3_SMALL=0 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash profiles/ab_env.sh "--steps 20 --warmup 5" "FHX_K3_SMALL=0" "FHX_K3_SMALL=1" "FHX_K3_SMALL=0" "FHX_K3_SMALL=1" > gpurun_out/r04/f_small_ab.txt 2>&1
bash profiles/ab_env.sh "--steps 40 --warmup 5 --max-chroms 3" "FHX_K3_SMALL=0" "FHX_K3_SMALL=1" >> gpurun_out/r04/f_small_ab.txt 2>&1
bash profiles/ab_env.sh "--steps 10 --warmup 3 --overdispersion 1.0" "FHX_K3_SMALL=1" >> gpurun_out/r04/f_small_ab.txt 2>&1
cat gpurun_out/r04/f_small_ab.txt
bash profiles/run_profile.sh r04/f_c3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/f_c3_profile.log 2>&1
head -24 gpurun_out/r04/f_c3_kernel_stats.txt
