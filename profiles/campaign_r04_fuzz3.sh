#!/bin/bash
# third fuzz campaign of round 4, on the final tree (fresh seeds): engine, large counts, sharded schedule, CLI end to end
mkdir -p gpurun_out/r04
{
echo "== test_gpu_fuzz 9000:9800"; FHX_FUZZ_SEEDS=9000:9800 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_native_dist 800:900"; FHX_FUZZ_SEEDS=800:900 timeout 1200 python -m pytest tests/test_gpu_native_dist.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_fuzz 102000:102150 (large counts)"; FHX_FUZZ_SEEDS=102000:102150 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
} > gpurun_out/r04/zz_fuzz_campaign.txt 2>&1
cat gpurun_out/r04/zz_fuzz_campaign.txt
