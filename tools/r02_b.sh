#!/bin/bash
mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm_gen3.py tests/test_gpu_unet.py -q -m gpu -x > gpurun_out/r02b/pytest.log 2>&1
tail -5 gpurun_out/r02b/pytest.log
for mode in split fp16; do
EW_BENCH_FULL_BREAKDOWN=1 EW_RESIDUAL=$mode python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline > gpurun_out/r02b/bench_$mode.log 2>&1
grep "ms  avg\|sum of" gpurun_out/r02b/bench_$mode.log | head -16
grep -o '"unet_forward_ms": [0-9.]*' gpurun_out/r02b/bench_$mode.log
done
