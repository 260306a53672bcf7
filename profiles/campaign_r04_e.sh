#!/bin/bash
mkdir -p gpurun_out/r04
for v in 0 1 2; do
FHX_RS_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r04/e_tests_$v.txt 2>&1
tail -3 gpurun_out/r04/e_tests_$v.txt | cut -c1-300
done
bash profiles/ab_env.sh "--steps 10 --warmup 3 --overdispersion 1.0" "FHX_RS_VARIANT=0" "FHX_RS_VARIANT=1" "FHX_RS_VARIANT=2" "FHX_RS_VARIANT=0" "FHX_RS_VARIANT=1" > gpurun_out/r04/e_rs_ab.txt 2>&1
cat gpurun_out/r04/e_rs_ab.txt
FHX_RS_VARIANT=1 bash profiles/run_profile.sh r04/e_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline --no-parity-check > gpurun_out/r04/e_od1_profile.log 2>&1
head -20 gpurun_out/r04/e_od1_kernel_stats.txt
