#!/bin/bash
# fuzz campaign on the final tree of round 4 (differential against the oracle): engine, sharded schedule, text ingest, writer
mkdir -p gpurun_out/r04
{
echo "== test_gpu_bench"; timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_fuzz 5000:5600"; FHX_FUZZ_SEEDS=5000:5600 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_fuzz 100000:100300 (large counts)"; FHX_FUZZ_SEEDS=100000:100300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_native_dist 400:520"; FHX_FUZZ_SEEDS=400:520 timeout 1200 python -m pytest tests/test_gpu_native_dist.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_ingest / writer / inflate 0:150"; FHX_FUZZ_SEEDS=0:150 timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_writer.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -2
} > gpurun_out/r04/p_fuzz_campaign.txt 2>&1
cat gpurun_out/r04/p_fuzz_campaign.txt
