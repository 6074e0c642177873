#!/bin/bash
# reprojection stage: bench + rocprofv3 kernel trace of the same command
mkdir -p gpurun_out/r02c
python bench_reproject.py --iters 5 > gpurun_out/r02c/bench_reproject.json 2> gpurun_out/r02c/bench_reproject.log
cat gpurun_out/r02c/bench_reproject.log | tail -14
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02c/prof -o reproj -- python $GRAFT_REPO_ROOT/bench_reproject.py --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/r02c/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r02c/prof -name "*.db" | head -2
DB=$(find gpurun_out/r02c/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r02c/reproject_kernel_stats.md
find gpurun_out/r02c/prof -name "*stats*" | head
head -30 gpurun_out/r02c/reproject_kernel_stats.md
