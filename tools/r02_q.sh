#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O
bash tools/r02_g.sh n > $O/r02g.log 2>&1; tail -32 $O/r02g.log | cut -c1-170
bash tools/pmc_traffic.sh > $O/traffic.log 2>&1; tail -1 $O/traffic.log | cut -c1-700
