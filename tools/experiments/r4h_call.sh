cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4h
for g in 3 1 2; do
EW_GEMM_GEN=$g EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 3 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4h/breakdown_gen$g.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GEN=$g forward ms', d['config']['unet_forward_ms'])"
done
