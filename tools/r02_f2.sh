#!/bin/bash
python tools/sk_bench.py 2>&1 | grep "SK=" | head -1
for a in 1 4 8 12 13; do echo "ablate $a:"; EW_LIB_PATH=$GRAFT_REPO_ROOT/evoworld_amd/libevoworld_hip_skab$a.so python tools/sk_bench.py 2>&1 | grep "SK=" | head -1; done
