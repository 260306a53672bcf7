#!/bin/bash
# second fuzz campaign of round 4, on the tree whose class kernels read their queues as dense lists (fresh seeds)
mkdir -p gpurun_out/r04
{
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "dense_lists or kernel_seconds or homogeneous" 2>&1 | tail -2
echo "== test_gpu_bench"; timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_fuzz 7000:7800"; FHX_FUZZ_SEEDS=7000:7800 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_fuzz 101000:101300 (large counts)"; FHX_FUZZ_SEEDS=101000:101300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_native_dist 600:700"; FHX_FUZZ_SEEDS=600:700 timeout 1200 python -m pytest tests/test_gpu_native_dist.py -x -q -m gpu 2>&1 | tail -2
} > gpurun_out/r04/u_fuzz_campaign.txt 2>&1
cat gpurun_out/r04/u_fuzz_campaign.txt
