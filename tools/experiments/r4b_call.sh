cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4b
EW_BENCH_FULL_BREAKDOWN=1 EW_BENCH_BY_SHAPE=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream > gpurun_out/r4b/bench_by_shape.log 2>&1
tail -5 gpurun_out/r4b/bench_by_shape.log
