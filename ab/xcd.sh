#!/bin/bash
ROOT=$PWD
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
for v in off on off on; do
  unset FHX_CL_XCD
  if [ $v = on ]; then export FHX_CL_XCD=1; fi
  rm -rf /tmp/prof_ab
  FHX_BENCH_HASH=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o run -- python $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity-check 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('xcd $v: pass %.3f ms  %s digest %s' % (d['ms_per_step'], {k: round(v,3) for k,v in d['kernels_ms'].items()}, d.get('result_digest')))"
  DB=$(find /tmp/prof_ab -name '*.db' | head -1)
  python $ROOT/profiles/summarize_rocprof.py $DB | sed -n 3,12p
done
