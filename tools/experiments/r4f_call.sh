cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py tests/test_gpu_pipeline.py -m gpu -x -q -s -k "streamk or cfg" 2>&1 | tail -15 > gpurun_out/r4f/pytest.log; cat gpurun_out/r4f/pytest.log
for mk in 0 2560 1280 0 2560; do
EW_G3_SKHALF_MINK=$mk EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4f/breakdown_mk$mk.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SKHALF_MINK=$mk forward ms', d['config']['unet_forward_ms'])"
done > gpurun_out/r4f/ab_skhalf.txt
cat gpurun_out/r4f/ab_skhalf.txt
