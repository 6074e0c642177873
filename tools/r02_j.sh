#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02j; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
tail -5 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
