#!/bin/bash
# current build: default bench (with breakdown) + by-shape breakdown + rocprofv3 kernel stats of a short run.  Usage: tools/profile_forward.sh <tag>
TAG=${1:-g}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
EW_BENCH_FULL_BREAKDOWN=1 python $R/bench.py ${BENCH_ARGS:-} > $O/bench.log 2>&1
grep '^{' $O/bench.log | tail -1 > $O/bench.json
grep -v '^{' $O/bench.log | tail -40
python -c "
import json
d = json.load(open('$O/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step')}, d['config']['unet_forward_ms'], d['roofline']['frac'], d.get('cpu_baseline'))"
EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 python $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline 2>&1 | grep -v '^{' > $O/by_shape.log
head -60 $O/by_shape.log
rocprofv3 --kernel-trace --stats -d $O/prof -o fwd -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-fp16-stream > $O/rocprof.log 2>&1
DB=$(find $O/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_stats.py $DB > $O/kernel_stats.md; fi
rm -rf $O/prof
head -30 $O/kernel_stats.md | cut -c1-160
