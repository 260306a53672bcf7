#!/bin/bash
mkdir -p gpurun_out/r04
python profiles/scaling_model.py --config C3 --steps 40 > gpurun_out/r04/g_scaling_model.txt 2>&1
cat gpurun_out/r04/g_scaling_model.txt
bash profiles/run_profile.sh r04/g_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline --no-parity-check > gpurun_out/r04/g_od1_profile.log 2>&1
head -22 gpurun_out/r04/g_od1_kernel_stats.txt
timeout 900 python -m pytest tests/test_gpu_native_dist.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -3
