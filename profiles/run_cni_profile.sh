#!/bin/bash
# usage: profiles/run_cni_profile.sh <tag>   -> gpurun_out/<tag>_cni_bench.json, <tag>_cni_kernel_stats.txt
set -u
TAG=${1:-cni}
REPO=$(pwd)
mkdir -p gpurun_out
python bench.py --path cni > gpurun_out/${TAG}_cni_bench.json 2> gpurun_out/${TAG}_cni_bench.err
tail -3 gpurun_out/${TAG}_cni_bench.err
cat gpurun_out/${TAG}_cni_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cniprof
rocprofv3 --kernel-trace --stats -d /tmp/cniprof -o run -- python $REPO/profiles/cni_bench.py > /tmp/cniprof.log 2>&1
cd $REPO
DB=$(find /tmp/cniprof -name "*.db" | head -1)
python profiles/summarize_rocprof.py "$DB" > gpurun_out/${TAG}_cni_kernel_stats.txt 2>&1
head -30 gpurun_out/${TAG}_cni_kernel_stats.txt
