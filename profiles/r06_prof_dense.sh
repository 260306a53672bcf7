#!/bin/bash
# per-kernel times of K3 with and without the dense q array, at 39 % and 55 % survivors and on the headline
for v in 0 1; do
  export FHX_K3_DENSE=$v
  bash profiles/run_profile.sh r06/c_hot39_dense$v --hotspots 0.2:4.5 --steps 5 --warmup 1 --no-cpu-baseline --no-parity-check --no-k3-stress > /dev/null 2>&1
  bash profiles/run_profile.sh r06/c_hot55_dense$v --hotspots 0.25:3.9 --steps 5 --warmup 1 --no-cpu-baseline --no-parity-check --no-k3-stress > /dev/null 2>&1
  bash profiles/run_profile.sh r06/c_c3_dense$v --steps 5 --warmup 1 --no-cpu-baseline --no-parity-check --no-k3-stress > /dev/null 2>&1
done
for f in gpurun_out/r06/c_*_kernel_stats.txt; do echo "== $f"; grep -E "k3_|bh_|os_|fill" $f | cut -c1-150; done
