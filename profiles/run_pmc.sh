#!/bin/bash
# Usage (GPU box, repo root): bash profiles/run_pmc.sh <tag> [bench args...]
# Two separate counter passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
set -e
TAG=$1; shift
T=${TAG//\//_}            # the tag may name a sub-directory of gpurun_out/; /tmp paths use a flat name
mkdir -p "$(dirname "$(pwd)/gpurun_out/${TAG}_x")"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${T}_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${T}_$C -o run -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-k3-stress > /tmp/pmc_${T}_$C.log 2>&1 || tail -5 /tmp/pmc_${T}_$C.log
done
python "$ROOT/profiles/summarize_pmc.py" $(find /tmp/pmc_${T}_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_${T}_WRITE_SIZE -name '*.db' | head -1) > "$ROOT/gpurun_out/${TAG}_pmc.txt"
cat "$ROOT/gpurun_out/${TAG}_pmc.txt"
