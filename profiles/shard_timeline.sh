#!/bin/bash
# every dispatch of one pass over the largest shard of an N-way sharding of C3 (default N = 8), with the gap before it:
# what the per-pass fixed cost of a strong-scaling run is made of.   bash profiles/shard_timeline.sh [N] > gpurun_out/...
N=${1:-8}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -o run -- python "$ROOT/bench.py" --shard-of $N --steps 5 --warmup 2 --no-cpu-baseline --no-parity-check --no-k3-stress > /tmp/prof_tl.log 2>&1
python "$ROOT/profiles/pass_timeline.py" "$(find /tmp/prof_tl -name '*.db' | head -1)"
