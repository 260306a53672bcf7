#!/bin/bash
# usage: profiles/run_kr_profile.sh <tag>   -> gpurun_out/<tag>_kr_bench.json, <tag>_kr_kernel_stats.txt
set -u
TAG=${1:-kr}
REPO=$(pwd)
mkdir -p gpurun_out
python bench.py --path kr > gpurun_out/${TAG}_kr_bench.json 2> gpurun_out/${TAG}_kr_bench.err
tail -3 gpurun_out/${TAG}_kr_bench.err
cat gpurun_out/${TAG}_kr_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/krprof
rocprofv3 --kernel-trace --stats -d /tmp/krprof -o run -- python $REPO/profiles/kr_bench.py > /tmp/krprof.log 2>&1
cd $REPO
DB=$(find /tmp/krprof -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$DB" > gpurun_out/${TAG}_kr_kernel_stats.txt 2>&1
head -30 gpurun_out/${TAG}_kr_kernel_stats.txt
