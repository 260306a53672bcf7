mkdir -p gpurun_out/r03
{
echo "== all GPU tests"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
echo "== test_gpu_fuzz 3000:3400"; FHX_FUZZ_SEEDS=3000:3400 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
echo "== test_gpu_native_dist (2-3 ranks over the pipe transport; thirds of the cases -r 0 / off the grid) 100:260"; FHX_FUZZ_SEEDS=100:260 timeout 900 python -m pytest tests/test_gpu_native_dist.py -x -q -m gpu 2>&1 | tail -2
} > gpurun_out/r03/p_campaign.txt 2>&1
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r03/p_c2_bench.json 2> gpurun_out/r03/p_c2_bench.err
cat gpurun_out/r03/p_campaign.txt; tail -c 1500 gpurun_out/r03/p_c2_bench.json
