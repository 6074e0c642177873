cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
R=$GRAFT_REPO_ROOT
for L in "" $R/evoworld_amd/libevoworld_hip_pff.so $R/evoworld_amd/libevoworld_hip_pg3.so "" $R/evoworld_amd/libevoworld_hip_pff.so $R/evoworld_amd/libevoworld_hip_pg3.so; do
n=$(basename "${L:-base}" .so)
EW_LIB_PATH=$L EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4s/bd_$n.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n forward ms', d['config']['unet_forward_ms'])"
grep "ff320" gpurun_out/r4s/bd_$n.txt
done > gpurun_out/r4s/ab.txt; cat gpurun_out/r4s/ab.txt
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s+(.*?)\s+n=\s*(\d+) total\s+([\d.]+) ms", l)
        if m: d[m.group(1).strip()]=(int(m.group(2)), float(m.group(3)))
    return d
a,b=load("gpurun_out/r4s/bd_base.txt"),load("gpurun_out/r4s/bd_libevoworld_hip_pg3.txt")
tot=0
for k in sorted(a, key=lambda k:-a[k][1]):
    if k in b and k.startswith("gemm3") and abs(a[k][1]-b[k][1])>0.03:
        print(f"{k:62s} n={a[k][0]:3d} {a[k][1]:7.2f} -> {b[k][1]:7.2f}  ({(b[k][1]-a[k][1]):+.2f} ms)")
    if k in b and k.startswith("gemm3"): tot+=b[k][1]-a[k][1]
print('gemm3 total delta', round(tot,2))
PY
