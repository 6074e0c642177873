#!/bin/bash
# A/B of two environment settings of the SAME library in one gpurun call (alternating processes on one box):
#   tools/ab_env.sh "EW_SPLIT_OPERANDS=0" "EW_SPLIT_OPERANDS=1"     (forward time + per-shape deltas of the launches)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_env; mkdir -p $O
for s in old new old new; do
E="$1"; [ $s = new ] && E="$2"
env $E EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 timeout 900 python $R/bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream > $O/bench_$s.log 2>&1
grep '^{' $O/bench_$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s ($E) forward ms', d['config']['unet_forward_ms'])"
done
python - <<PY
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s+(.*?)\s+n=\s*(\d+) total\s+([\d.]+) ms", l)
        if m: d[m.group(1).strip()]=(int(m.group(2)), float(m.group(3)))
    return d
a,b=load("$O/bench_old.log"),load("$O/bench_new.log")
tot=0
for k in sorted(set(a)|set(b)):
    x,y=a.get(k,(0,0.0)),b.get(k,(0,0.0))
    if abs(x[1]-y[1])>0.05:
        print(f"{k:62s} n={x[0]:3d}/{y[0]:3d} {x[1]:7.2f} -> {y[1]:7.2f}  ({(y[1]-x[1]):+.2f} ms)")
    tot+=y[1]-x[1]
print('total delta', round(tot,2))
PY
