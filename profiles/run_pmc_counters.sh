#!/bin/bash
# Usage (GPU box, repo root): bash profiles/run_pmc_counters.sh <tag> COUNTER [COUNTER...]
# One rocprofv3 --pmc pass per counter (kernel-trace only), summed per kernel: total and per launch.
set -e
TAG=$1; shift
T=${TAG//\//_}            # the tag may name a sub-directory of gpurun_out/; /tmp paths use a flat name
mkdir -p "$(dirname "$(pwd)/gpurun_out/${TAG}_x")"
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
: > "$ROOT/gpurun_out/${TAG}_counters.txt"
for C in "$@"; do
  rm -rf /tmp/pmcc_${T}_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcc_${T}_$C -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-k3-stress --steps 3 > /tmp/pmcc_${T}_$C.log 2>&1 || { echo "$C: failed"; tail -3 /tmp/pmcc_${T}_$C.log; continue; }
  DB=$(find /tmp/pmcc_${T}_$C -name '*.db' | head -1)
  python - "$DB" "$C" >> "$ROOT/gpurun_out/${TAG}_counters.txt" <<'PY'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for name, counter, total, n in rows:
    if "k2h_" in name or "k2_queue" in name or "k1_classify" in name or "k2_classify" in name or "k3_compact" in name:
        m = re.search(r"(k\d\w*_\w+(<[^>]*>)?)", name)
        short = re.sub(r"\(fhx::dev::BranchClass\)", "", m.group(1)) if m else name[:40]
        print("%-14s %-40s launches %3d total %.6g per-launch %.6g" % (counter, short, n, total, total / n))
PY
done
cat "$ROOT/gpurun_out/${TAG}_counters.txt"
