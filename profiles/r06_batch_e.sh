#!/bin/bash
# (working script of round 6, kept because evidence files name it; FHX_EMIT_WRITERS was the knob of the copy-out experiment and left with it: profiles/r06/cli_c5.txt)
mkdir -p gpurun_out/r06
(timeout 1200 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -30) > gpurun_out/r06/e_gputests.txt
{
echo "== C3 headline";        bash profiles/ab_env.sh "--steps 10 --warmup 2" "FHX_Q_PREFILL=0" "FHX_X=default"
echo "== lognormal s = 1.0";  bash profiles/ab_env.sh "--overdispersion 1.0 --steps 8 --warmup 2" "FHX_Q_PREFILL=0" "FHX_X=default"
echo "== hotspots 0.25:3.9";  bash profiles/ab_env.sh "--hotspots 0.25:3.9 --steps 8 --warmup 2" "FHX_Q_PREFILL=0" "FHX_X=default"
echo "== 1/8 shard";          bash profiles/ab_env.sh "--shard-of 8 --steps 40 --warmup 5" "FHX_Q_PREFILL=0" "FHX_X=default"
echo "== C2";                 bash profiles/ab_env.sh "--config C2 --steps 20 --warmup 3" "FHX_Q_PREFILL=0" "FHX_X=default"
} > gpurun_out/r06/e_ab_prefill.txt 2>&1
bash profiles/shard_timeline.sh 8 > gpurun_out/r06/e_tl_shard8.txt 2>&1
python profiles/scaling_model.py --steps 40 > gpurun_out/r06/e_scaling_model.txt 2>&1
python bench.py --config C2 --steps 20 --warmup 3 --no-cpu-baseline --no-parity-check --no-k3-stress > gpurun_out/r06/e_c2_bench.json 2>/dev/null
python profiles/cli_c5.py --again-with FHX_EMIT_WRITERS=1 --again-with FHX_EMIT_WRITERS=3 --again-with FHX_EMIT_WRITERS=12 > gpurun_out/r06/e_cli_c5.txt 2>&1
tail -5 gpurun_out/r06/e_gputests.txt; cut -c1-130 gpurun_out/r06/e_ab_prefill.txt; grep -v "^W2026\|^E2026" gpurun_out/r06/e_tl_shard8.txt | head -40; cat gpurun_out/r06/e_scaling_model.txt; python -c "
import json; d=json.load(open('gpurun_out/r06/e_c2_bench.json')); print('C2', d['ms_per_step'], d.get('ms_per_pass'), d['kernels_ms'])"; cat gpurun_out/r06/e_cli_c5.txt
