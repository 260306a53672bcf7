#!/bin/bash
# Usage (GPU box, repo root): bash profiles/run_pmc.sh <tag> [bench args...]
# Two separate counter passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
set -e
TAG=$1; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${TAG}_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${TAG}_$C -o run -- python "$ROOT/bench.py" "$@" --no-cpu-baseline > /tmp/pmc_${TAG}_$C.log 2>&1 || tail -5 /tmp/pmc_${TAG}_$C.log
done
python "$ROOT/profiles/summarize_pmc.py" $(find /tmp/pmc_${TAG}_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_${TAG}_WRITE_SIZE -name '*.db' | head -1) > "$ROOT/gpurun_out/${TAG}_pmc.txt"
cat "$ROOT/gpurun_out/${TAG}_pmc.txt"
