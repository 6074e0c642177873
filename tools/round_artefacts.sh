#!/bin/bash
# round-2 artefacts of the current build: full GPU suite, default bench, kernel stats, HBM traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
bash tools/profile_forward.sh n > $O/r02g.log 2>&1; tail -45 $O/r02g.log | cut -c1-170
bash tools/pmc_traffic.sh > $O/traffic.log 2>&1; tail -3 $O/traffic.log | cut -c1-600
