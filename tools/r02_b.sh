#!/bin/bash
mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_unet.py tests/test_gpu_pipeline.py -q -m gpu -s -k "tiny or 25_steps" 2>&1 | grep "rel-L2\|passed\|failed" | grep -v tap
EW_BENCH_FULL_BREAKDOWN=1 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline > gpurun_out/r02b/bench.log 2>&1
grep "ms  avg\|sum of" gpurun_out/r02b/bench.log | head -12
grep -o '"unet_forward_ms": [0-9.]*' gpurun_out/r02b/bench.log
