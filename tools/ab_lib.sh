#!/bin/bash
# A/B of two builds of the library in one gpurun call: tools/ab_lib.sh <other.so>  (forward time + per-shape deltas of the GEMMs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_lib; mkdir -p $O
for s in old new old new; do
L=""; [ $s = old ] && L=$R/$1
EW_LIB_PATH=$L EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream > $O/bench_$s.log 2>&1
grep '^{' $O/bench_$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s forward ms', d['config']['unet_forward_ms'])"
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
for k in a:
    if k in b and abs(a[k][1]-b[k][1])>0.05:
        print(f"{k:62s} n={a[k][0]:3d} {a[k][1]:7.2f} -> {b[k][1]:7.2f}  ({(b[k][1]-a[k][1]):+.2f} ms)")
    if k in b: tot+=b[k][1]-a[k][1]
print('total delta', round(tot,2))
PY
