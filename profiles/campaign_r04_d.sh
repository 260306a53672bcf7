#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r04/d_tests.txt 2>&1
tail -5 gpurun_out/r04/d_tests.txt
bash profiles/ab_env.sh "--steps 20 --warmup 5" "FHX_CL_BASE=1" "FHX_CL_PACK=0 FHX_CL_TB=0" "FHX_CL_PACK=1 FHX_CL_TB=0" "FHX_CL_PACK=0 FHX_CL_TB=1" "FHX_CL_PACK=1 FHX_CL_TB=1" "FHX_CL_BASE=1" "FHX_CL_PACK=1 FHX_CL_TB=1" > gpurun_out/r04/d_classify_ab.txt 2>&1
cat gpurun_out/r04/d_classify_ab.txt
