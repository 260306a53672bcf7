#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04/c_tests.txt
bash profiles/ab_env.sh "--steps 10 --warmup 3 --overdispersion 1.0" "FHX_RS_WAVES=4" "FHX_RS_WAVES=6" "FHX_RS_WAVES=4" "FHX_RS_WAVES=6" > gpurun_out/r04/c_rs_ab.txt 2>&1
bash profiles/run_profile.sh r04/c_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline --no-parity-check > gpurun_out/r04/c_od1_profile.log 2>&1
cat gpurun_out/r04/c_tests.txt gpurun_out/r04/c_rs_ab.txt; head -24 gpurun_out/r04/c_od1_kernel_stats.txt
