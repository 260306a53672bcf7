#!/bin/bash
ROOT=$PWD
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o run -- python $ROOT/bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C2: step %.3f ms  %s per pass %s' % (d['ms_per_step'], {k: round(v,3) for k,v in d['kernels_ms'].items()}, d.get('ms_per_pass')))"
DB=$(find /tmp/prof_c2 -name '*.db' | head -1)
python $ROOT/profiles/pass_timeline.py $DB
