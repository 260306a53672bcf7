#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [bench args...]
# Writes gpurun_out/<tag>_kernel_stats.txt (rocprofv3 --kernel-trace --stats, condensed) and gpurun_out/<tag>_bench.json
set -e
TAG=$1; shift
T=${TAG//\//_}            # the tag may name a sub-directory of gpurun_out/; /tmp paths use a flat name
mkdir -p "$(dirname "$(pwd)/gpurun_out/${TAG}_x")"
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
python bench.py "$@" > "$ROOT/gpurun_out/${TAG}_bench.json" 2> "$ROOT/gpurun_out/${TAG}_bench.err"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$T
rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o run -- python "$ROOT/bench.py" "$@" --no-cpu-baseline --no-k3-stress > /tmp/prof_$T.log 2>&1
DB=$(find /tmp/prof_$T -name '*.db' | head -1)
python "$ROOT/profiles/summarize_rocprof.py" "$DB" > "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
cat "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
python -c "import json; d=json.load(open('$ROOT/gpurun_out/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
