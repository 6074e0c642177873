cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4d
for m in 1 0 1 0; do
EW_ATTN_LOG2=$m timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline --no-fp16-stream 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('EW_ATTN_LOG2=$m forward ms', d['config']['unet_forward_ms'])"
done > gpurun_out/r4d/ab_attn_log2.txt
cat gpurun_out/r4d/ab_attn_log2.txt
