#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O
for s in 0 2 0 2; do
EW_G3_SHORT=$s EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline > $O/bench$s.log 2>&1
grep '^{' $O/bench$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('EW_G3_SHORT=$s forward ms', d['config']['unet_forward_ms'])"
grep "M=460800 N=320 K=1280" $O/bench$s.log
done
