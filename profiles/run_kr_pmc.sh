#!/bin/bash
# Usage (GPU box, repo root): bash profiles/run_kr_pmc.sh <tag>
# HBM bytes of the Knight-Ruiz kernels: two separate counter passes with --kernel-trace only (MI355X_MICROARCH.md).
set -e
TAG=$1
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/krpmc_${TAG}_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/krpmc_${TAG}_$C -o run -- python "$ROOT/profiles/kr_bench.py" > /tmp/krpmc_${TAG}_$C.log 2>&1 || tail -5 /tmp/krpmc_${TAG}_$C.log
done
python "$ROOT/profiles/summarize_pmc.py" $(find /tmp/krpmc_${TAG}_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/krpmc_${TAG}_WRITE_SIZE -name '*.db' | head -1) > "$ROOT/gpurun_out/${TAG}_kr_pmc.txt"
cat "$ROOT/gpurun_out/${TAG}_kr_pmc.txt"
