#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O
for s in 0 1; do
EW_G3_SK=$s EW_G3_SK_MINK=${MINK:-640} EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline > $O/bench$s.log 2>&1
grep '^{' $O/bench$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('EW_G3_SK=$s forward ms', d['config']['unet_forward_ms'])"
done
python - <<PY
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s+(.*?)\s+n=\s*(\d+) total\s+([\d.]+) ms", l)
        if m: d[m.group(1).strip()]=(int(m.group(2)), float(m.group(3)))
    return d
a,b=load("$O/bench0.log"),load("$O/bench1.log")
tot=0
for k in a:
    if k in b and abs(a[k][1]-b[k][1])>0.03 and 'gemm3' in k:
        print(f"{k:60s} n={a[k][0]:3d} {a[k][1]:7.2f} -> {b[k][1]:7.2f}  ({(b[k][1]-a[k][1]):+.2f} ms)")
        tot+=b[k][1]-a[k][1]
print('total gemm3 delta', round(tot,2))
PY
