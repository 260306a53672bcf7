#!/bin/bash
# end-of-round evidence on the final tree (two gpurun calls: `A`, `B`)
mkdir -p gpurun_out/r06
if [ "$1" = "A" ]; then
  timeout 1100 bash profiles/run_profile.sh r06/z --steps 20 --warmup 5 > gpurun_out/r06/z_profile.log 2>&1
  timeout 1200 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r06/z_gpu_tests.txt 2>&1
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/z_smoke.txt 2>&1
  timeout 300 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r06/z_c2_bench.json 2> gpurun_out/r06/z_c2_bench.err
  timeout 900 python bench.py --config C3w --steps 5 --warmup 2 > gpurun_out/r06/z_c3w_bench.json 2> gpurun_out/r06/z_c3w_bench.err
  timeout 1200 python bench.py --config C5 --steps 3 --warmup 1 > gpurun_out/r06/z_c5_bench.json 2> gpurun_out/r06/z_c5_bench.err
  FHX_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-weak > gpurun_out/r06/z_forced_dist_bench.json 2> gpurun_out/r06/z_forced_dist_bench.err
  tail -12 gpurun_out/r06/z_gpu_tests.txt; tail -1 gpurun_out/r06/z_smoke.txt
  head -34 gpurun_out/r06/z_kernel_stats.txt
  for f in z z_c2 z_c3w z_c5 z_forced_dist; do python -c "
import json
d=json.loads(open('gpurun_out/r06/${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['kernels_ms'], (d.get('parity_check') or {}).get('ok'), d.get('strong_efficiency'), (d.get('totals_semantics') or {}).get('mode'), [(k.get('survivor_fraction'), k.get('k3_ms'), (k.get('parity_check') or {}).get('ok')) for k in d.get('k3_stress', [])])"; done
elif [ "$1" = "B" ]; then
  timeout 600 python profiles/scaling_model.py --config C3 --steps 40 > gpurun_out/r06/z_scaling_model.txt 2>&1
  timeout 400 bash profiles/run_profile.sh r06/z_od1 --steps 10 --warmup 3 --overdispersion 1.0 --no-cpu-baseline > gpurun_out/r06/z_od1_profile.log 2>&1
  timeout 400 bash profiles/run_profile.sh r06/z_hot39 --steps 10 --warmup 3 --hotspots 0.2:4.5 --no-cpu-baseline > gpurun_out/r06/z_hot39_profile.log 2>&1
  timeout 400 bash profiles/run_profile.sh r06/z_hot55 --steps 10 --warmup 3 --hotspots 0.25:3.9 --no-cpu-baseline > gpurun_out/r06/z_hot55_profile.log 2>&1
  timeout 1000 bash profiles/run_pmc.sh r06/z --steps 3 --warmup 1 --no-parity-check > gpurun_out/r06/z_pmc.log 2>&1
  timeout 1000 bash profiles/run_pmc.sh r06/z_hot55 --steps 3 --warmup 1 --hotspots 0.25:3.9 --no-parity-check > gpurun_out/r06/z_hot55_pmc.log 2>&1
  timeout 1500 bash profiles/run_pmc_counters.sh r06/z SQ_INSTS_VALU GRBM_GUI_ACTIVE > gpurun_out/r06/z_counters.log 2>&1
  bash profiles/shard_timeline.sh 8 > gpurun_out/r06/z_tl_shard8.txt 2>&1
  { echo "== test_gpu_k3_sort large-sort fuzz 30000:31000"; FHX_FUZZ_SEEDS=30000:31000 timeout 900 python -m pytest tests/test_gpu_k3_sort.py -x -q -k fuzz 2>&1 | tail -2
    echo "== test_gpu_fuzz 12000:12300"; FHX_FUZZ_SEEDS=12000:12300 timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
    echo "== test_gpu_native_dist 2000:2060"; FHX_FUZZ_SEEDS=2000:2060 timeout 600 python -m pytest tests/test_gpu_native_dist.py -x -q -m gpu 2>&1 | tail -2
    echo "== test_native_io (parts protocol fuzz) 100:160"; FHX_FUZZ_SEEDS=100:160 timeout 600 python -m pytest tests/test_native_io.py -x -q -k "random" 2>&1 | tail -2
  } > gpurun_out/r06/zz_fuzz_campaign.txt 2>&1
  tail -12 gpurun_out/r06/z_scaling_model.txt
  for t in z_od1 z_hot39 z_hot55; do echo "== $t"; grep -E "os_|bh_|k3_|ks_" gpurun_out/r06/${t}_kernel_stats.txt; done
  grep -E "kernel |k2h_heavy|k2_classify|k1_classify|k3_compact" gpurun_out/r06/z_pmc.txt
  grep -E "kernel |os_scatter|bh_apply|k3_compact|k3_fill" gpurun_out/r06/z_hot55_pmc.txt
  cat gpurun_out/r06/z_counters.txt | head -30
  cat gpurun_out/r06/zz_fuzz_campaign.txt
fi
