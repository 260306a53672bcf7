#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -2
for b in 8 9 10; do FHX_RS_BITS=$b timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bh" 2>&1 | tail -1; done
bash profiles/ab_env.sh "--steps 10 --warmup 3 --overdispersion 1.0" "FHX_RS_BITS=11" "FHX_RS_BITS=10" "FHX_RS_BITS=9" "FHX_RS_BITS=8" "FHX_RS_BITS=11" "FHX_RS_BITS=9" > gpurun_out/r04/n_rs_bits_ab.txt 2>&1
bash profiles/ab_env.sh "--steps 5 --warmup 2 --config C3w" "FHX_RS_BITS=11" "FHX_RS_BITS=9" "FHX_RS_BITS=8" >> gpurun_out/r04/n_rs_bits_ab.txt 2>&1
cat gpurun_out/r04/n_rs_bits_ab.txt
