#!/bin/bash
mkdir -p gpurun_out/r04
python profiles/scaling_model.py --config C3 --steps 40 > gpurun_out/r04/z_scaling_model.txt 2>&1
cat gpurun_out/r04/z_scaling_model.txt
bash profiles/run_profile.sh r04/z_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline > gpurun_out/r04/z_od1_profile.log 2>&1
head -22 gpurun_out/r04/z_od1_kernel_stats.txt
bash profiles/run_pmc.sh r04/z_od1 --steps 3 --warmup 1 --overdispersion 1.0 --no-parity-check > gpurun_out/r04/z_od1_pmc.log 2>&1
grep -E "rs_|bh_|k3_|kernel " gpurun_out/r04/z_od1_pmc.txt
