cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4o
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ff_fused.py tests/test_gpu_unet.py -m gpu -q -s -x -k "not full_size" 2>&1 | grep -v "^tap" | tail -25 > gpurun_out/r4o/pytest.log; cat gpurun_out/r4o/pytest.log
for L in "" $R/evoworld_amd/libevoworld_hip_ffb0.so "" $R/evoworld_amd/libevoworld_hip_ffb0.so; do
EW_LIB_PATH=$L EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4o/bd.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=${L:-default(bias in C)} forward ms', d['config']['unet_forward_ms'])"; grep ff320 gpurun_out/r4o/bd.txt
done > gpurun_out/r4o/ab.txt; cat gpurun_out/r4o/ab.txt
