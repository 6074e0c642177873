#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ab; mkdir -p $O
cd $R/_old_r01g && python $R/tools/ab_kernels.py > $O/old.log 2>&1
cd $R && python tools/ab_kernels.py > $O/new.log 2>&1
cd $R/_old_r01g && python $R/tools/ab_kernels.py > $O/old2.log 2>&1
paste $O/old.log $O/new.log $O/old2.log | cut -c1-200
