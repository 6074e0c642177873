#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o st -- python $R/tools/bench_stages.py > $O/stages.log 2>&1
grep "VAE\|CLIP\|peak" $O/stages.log
DB=$(find $O/prof -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kernel_stats.md; rm -rf $O/prof
head -40 $O/kernel_stats.md | cut -c1-150
