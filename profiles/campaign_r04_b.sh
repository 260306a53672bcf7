#!/bin/bash
mkdir -p gpurun_out/r04
bash profiles/run_profile.sh r04/b_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline --no-parity-check > gpurun_out/r04/b_od1_profile.log 2>&1
bash profiles/ab_env.sh "--steps 20 --warmup 5" "FHX_CL_HOIST=0" "FHX_CL_HOIST=1" "FHX_CL_HOIST=0" "FHX_CL_HOIST=1" > gpurun_out/r04/b_hoist_ab.txt 2>&1
FHX_CL_HOIST=1 bash profiles/run_profile.sh r04/b_hoist --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check > gpurun_out/r04/b_hoist_profile.log 2>&1
cat gpurun_out/r04/b_hoist_ab.txt; head -30 gpurun_out/r04/b_od1_kernel_stats.txt; head -12 gpurun_out/r04/b_hoist_kernel_stats.txt
