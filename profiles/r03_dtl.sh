mkdir -p gpurun_out/r03
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 FHX_FORCE_DIST=1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_dtl
rocprofv3 --kernel-trace -d /tmp/prof_dtl -o run -- python $GRAFT_REPO_ROOT/bench.py --max-chroms 1 --steps 3 --warmup 2 --no-cpu-baseline --no-parity-check --no-weak > /tmp/dtl.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/pass_timeline.py $(find /tmp/prof_dtl -name '*.db' | head -1) | tee gpurun_out/r03/m_dist_timeline.txt
tail -3 /tmp/dtl.log
