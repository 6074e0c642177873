#!/bin/bash
mkdir -p gpurun_out/r02b
EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 python bench.py --steps 1 --warmup 1 --denoise-steps 3 --no-cpu-baseline > gpurun_out/r02b/bench_shape.log 2>&1
grep "ms  avg\|sum of" gpurun_out/r02b/bench_shape.log
grep -o '"unet_forward_ms": [0-9.]*' gpurun_out/r02b/bench_shape.log
