cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4p
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_gemm_gen3.py tests/test_gpu_ops.py tests/test_gpu_vae.py tests/test_gpu_unet.py tests/test_gpu_fullshape_parity.py -m gpu -q -x -k "not full_size" 2>&1 | tail -6 > gpurun_out/r4p/pytest.log; cat gpurun_out/r4p/pytest.log
for L in "" $R/evoworld_amd/libevoworld_hip_g3old.so "" $R/evoworld_amd/libevoworld_hip_g3old.so; do
n=new; [ -n "$L" ] && n=old
EW_LIB_PATH=$L EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4p/bd_$n.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n forward ms', d['config']['unet_forward_ms'])"
done > gpurun_out/r4p/ab.txt; cat gpurun_out/r4p/ab.txt
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s+(.*?)\s+n=\s*(\d+) total\s+([\d.]+) ms", l)
        if m: d[m.group(1).strip()]=(int(m.group(2)), float(m.group(3)))
    return d
a,b=load("gpurun_out/r4p/bd_old.txt"),load("gpurun_out/r4p/bd_new.txt")
tot=0
for k in sorted(a, key=lambda k:-a[k][1]):
    if k in b and abs(a[k][1]-b[k][1])>0.04:
        print(f"{k:62s} n={a[k][0]:3d} {a[k][1]:7.2f} -> {b[k][1]:7.2f}  ({(b[k][1]-a[k][1]):+.2f} ms)")
    if k in b: tot+=b[k][1]-a[k][1]
print('total delta', round(tot,2))
PY
