cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4m
timeout 900 python -m pytest tests/test_gpu_ff_fused.py tests/test_gpu_unet.py -m gpu -q -s -x -k "folded or fused_feed_forward_modes or tiny_vs_oracle" 2>&1 | grep -v "^tap" | tail -14 > gpurun_out/r4m/pytest.log; cat gpurun_out/r4m/pytest.log
for m in 1 3 1 3; do
EW_FUSED_FF=$m EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4m/breakdown_ff$m.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('EW_FUSED_FF=$m forward ms', d['config']['unet_forward_ms'])"
done > gpurun_out/r4m/ab.txt; cat gpurun_out/r4m/ab.txt
grep -E "ff320|ln_kernel" gpurun_out/r4m/breakdown_ff1.txt gpurun_out/r4m/breakdown_ff3.txt
