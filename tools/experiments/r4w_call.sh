cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4w
R=$GRAFT_REPO_ROOT
EW_LIB_PATH=$R/evoworld_amd/libevoworld_hip_ed2.so timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py -m gpu -q -x 2>&1 | tail -2
EW_LIB_PATH=$R/evoworld_amd/libevoworld_hip_ed3.so timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py -m gpu -q -x 2>&1 | tail -2
for n in edold ed1 ed2 ed3 edold ed2 ed3; do
EW_LIB_PATH=$R/evoworld_amd/libevoworld_hip_$n.so EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4w/bd_$n.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n forward ms', d['config']['unet_forward_ms'])"
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s+(.*?)\s+n=\s*(\d+) total\s+([\d.]+) ms", l)
        if m: d[m.group(1).strip()]=(int(m.group(2)), float(m.group(3)))
    return d
a=load("gpurun_out/r4w/bd_edold.txt")
for n in ("ed2","ed3"):
    b=load(f"gpurun_out/r4w/bd_{n}.txt"); tot=0
    print("==",n)
    for k in sorted(a, key=lambda k:-a[k][1]):
        if k in b and k.startswith("gemm3") and abs(a[k][1]-b[k][1])>0.05:
            print(f"{k:62s} n={a[k][0]:3d} {a[k][1]:7.2f} -> {b[k][1]:7.2f}  ({(b[k][1]-a[k][1]):+.2f} ms)")
        if k in b and k.startswith("gemm3"): tot+=b[k][1]-a[k][1]
    print('gemm3 total delta', round(tot,2))
PY
