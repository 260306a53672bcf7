#!/bin/bash
# end-of-round evidence on the final tree (two gpurun calls: `A`, `B`)
mkdir -p gpurun_out/r05
if [ "$1" = "A" ]; then
  timeout 1100 bash profiles/run_profile.sh r05/z --steps 20 --warmup 5 > gpurun_out/r05/z_profile.log 2>&1
  timeout 900 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r05/z_gpu_tests.txt 2>&1
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/z_smoke.txt 2>&1
  timeout 300 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r05/z_c2_bench.json 2> gpurun_out/r05/z_c2_bench.err
  timeout 900 python bench.py --config C3w --steps 5 --warmup 2 > gpurun_out/r05/z_c3w_bench.json 2> gpurun_out/r05/z_c3w_bench.err
  timeout 1200 python bench.py --config C5 --steps 3 --warmup 1 > gpurun_out/r05/z_c5_bench.json 2> gpurun_out/r05/z_c5_bench.err
  FHX_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-weak > gpurun_out/r05/z_forced_dist_bench.json 2> gpurun_out/r05/z_forced_dist_bench.err
  tail -12 gpurun_out/r05/z_gpu_tests.txt; tail -1 gpurun_out/r05/z_smoke.txt
  head -30 gpurun_out/r05/z_kernel_stats.txt
  grep -E "^\[ *[0-9.]+ s\]" gpurun_out/r05/z_bench.err | cut -c1-160
  for f in z z_c2 z_c3w z_c5 z_forced_dist; do python -c "
import json
d=json.loads(open('gpurun_out/r05/${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['kernels_ms'], (d.get('parity_check') or {}).get('ok'), d.get('strong_efficiency'), [(k.get('survivor_fraction'), k.get('k3_ms'), (k.get('parity_check') or {}).get('ok')) for k in d.get('k3_stress', [])])"; done
elif [ "$1" = "B" ]; then
  timeout 600 python profiles/scaling_model.py --config C3 --steps 40 > gpurun_out/r05/z_scaling_model.txt 2>&1
  timeout 400 bash profiles/run_profile.sh r05/z_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline > gpurun_out/r05/z_od1_profile.log 2>&1
  timeout 400 bash profiles/run_profile.sh r05/z_hot39 --steps 10 --warmup 3 --hotspots 0.2:4.5 --no-cpu-baseline > gpurun_out/r05/z_hot39_profile.log 2>&1
  timeout 400 bash profiles/run_profile.sh r05/z_hot55 --steps 10 --warmup 3 --hotspots 0.25:3.9 --no-cpu-baseline > gpurun_out/r05/z_hot55_profile.log 2>&1
  timeout 1000 bash profiles/run_pmc.sh r05/z_hot39 --steps 3 --warmup 1 --hotspots 0.2:4.5 --no-parity-check > gpurun_out/r05/z_hot39_pmc.log 2>&1
  timeout 1500 bash profiles/run_pmc_counters.sh r05/z SQ_INSTS_VALU GRBM_GUI_ACTIVE > gpurun_out/r05/z_counters.log 2>&1
  { echo "== test_gpu_k3_sort large-sort fuzz 20000:21500"; FHX_FUZZ_SEEDS=20000:21500 timeout 900 python -m pytest tests/test_gpu_k3_sort.py -x -q -k fuzz 2>&1 | tail -2
    echo "== test_gpu_fuzz 11000:11300"; FHX_FUZZ_SEEDS=11000:11300 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
    echo "== test_gpu_native_dist 1000:1060"; FHX_FUZZ_SEEDS=1000:1060 timeout 600 python -m pytest tests/test_gpu_native_dist.py -x -q -m gpu 2>&1 | tail -2
  } > gpurun_out/r05/zz_fuzz_campaign.txt 2>&1
  tail -25 gpurun_out/r05/z_scaling_model.txt
  for t in z_od1 z_hot39 z_hot55; do echo "== $t"; grep -E "os_|bh_|k3_|ks_" gpurun_out/r05/${t}_kernel_stats.txt; done
  grep -E "kernel |os_|bh_apply|k3_compact" gpurun_out/r05/z_hot39_pmc.txt
  cat gpurun_out/r05/zz_fuzz_campaign.txt
fi
# `C`: the tree after the one-sweep tiles grew to 12 288 keys - bench line + kernel summary, all GPU tests, K3 at the three survivor
# fractions, PMC traffic at 39 %  (python bench.py ... ; bash profiles/campaign_r05_final.sh C)
if [ "$1" = "C" ]; then
  timeout 900 bash profiles/run_profile.sh r05/y --steps 20 --warmup 5 > gpurun_out/r05/y_profile.log 2>&1
  timeout 900 python -m pytest tests -x -q -m gpu --durations=4 > gpurun_out/r05/y_gpu_tests.txt 2>&1
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/y_smoke.txt 2>&1
  timeout 400 bash profiles/run_profile.sh r05/y_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline > gpurun_out/r05/y_od1_profile.log 2>&1
  timeout 400 bash profiles/run_profile.sh r05/y_hot39 --steps 10 --warmup 3 --hotspots 0.2:4.5 --no-cpu-baseline > gpurun_out/r05/y_hot39_profile.log 2>&1
  timeout 400 bash profiles/run_profile.sh r05/y_hot55 --steps 10 --warmup 3 --hotspots 0.25:3.9 --no-cpu-baseline > gpurun_out/r05/y_hot55_profile.log 2>&1
  timeout 1000 bash profiles/run_pmc.sh r05/y_hot39 --steps 3 --warmup 1 --hotspots 0.2:4.5 --no-parity-check > gpurun_out/r05/y_hot39_pmc.log 2>&1
  tail -8 gpurun_out/r05/y_gpu_tests.txt; tail -1 gpurun_out/r05/y_smoke.txt
  for t in y y_od1 y_hot39 y_hot55; do echo "== $t"; grep -E "os_|bh_|k3_|ks_|k2h_heavy" gpurun_out/r05/${t}_kernel_stats.txt; python -c "
import json
d=json.loads(open('gpurun_out/r05/${t}_bench.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], d['kernels_ms'], (d.get('parity_check') or {}).get('ok'), d.get('bh_rows_sorted_rank0'), [(k.get('survivor_fraction'), k.get('k3_ms'), k.get('sorted_keys_per_s'), (k.get('parity_check') or {}).get('ok')) for k in d.get('k3_stress', [])])"; done
  grep -E "kernel |os_scatter|bh_apply|k3_compact|os_hist|os_find" gpurun_out/r05/y_hot39_pmc.txt
fi
