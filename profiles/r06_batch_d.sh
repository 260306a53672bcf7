#!/bin/bash
# (working script of round 6, kept because evidence files name it)
mkdir -p gpurun_out/r06
(timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -30) > gpurun_out/r06/d_gputests.txt
B="--steps 8 --warmup 2"
{
echo "== C3 headline";        bash profiles/ab_env.sh "$B" "FHX_K3_DENSE=0" "FHX_X=default"
echo "== lognormal s = 1.0";  bash profiles/ab_env.sh "--overdispersion 1.0 $B" "FHX_K3_DENSE=0" "FHX_X=default"
echo "== hotspots 0.2:4.5";   bash profiles/ab_env.sh "--hotspots 0.2:4.5 $B" "FHX_K3_DENSE=0" "FHX_X=default"
echo "== hotspots 0.25:3.9";  bash profiles/ab_env.sh "--hotspots 0.25:3.9 $B" "FHX_K3_DENSE=0" "FHX_X=default"
echo "== 1/8 shard";          bash profiles/ab_env.sh "--shard-of 8 --steps 30 --warmup 5" "FHX_K3_DENSE=0" "FHX_X=default"
echo "== C2 (FHX_FIT_SERIAL=1: per-count tables on the fitting thread)"; bash profiles/ab_env.sh "--config C2 --steps 20 --warmup 3" "FHX_FIT_SERIAL=1" "FHX_X=default"
} > gpurun_out/r06/d_ab.txt 2>&1
FHX_FIT_TIMES=1 python bench.py --config C2 --steps 4 --warmup 2 --no-cpu-baseline --no-parity-check --no-k3-stress 2>&1 >/dev/null | grep -v "^\[" | tail -4 > gpurun_out/r06/d_c2_fit_times.txt
bash profiles/r06_cli_plain_gzip.sh > gpurun_out/r06/d_cli_plain_gzip.txt 2>&1
tail -5 gpurun_out/r06/d_gputests.txt; cat gpurun_out/r06/d_ab.txt | cut -c1-130; cat gpurun_out/r06/d_c2_fit_times.txt; cat gpurun_out/r06/d_cli_plain_gzip.txt
