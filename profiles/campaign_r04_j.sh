#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/j_gpu_tests.txt 2>&1
tail -4 gpurun_out/r04/j_gpu_tests.txt | cut -c1-300
bash profiles/ab_env.sh "--steps 20 --warmup 5" "FHX_X=1" "FHX_X=2" > gpurun_out/r04/j_digest.txt 2>&1
cat gpurun_out/r04/j_digest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
