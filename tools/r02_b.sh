#!/bin/bash
mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_pipeline.py tests/test_gpu_fullshape_parity.py -q -m gpu -s -k "groupnorm or layernorm or unet or denoise or call_surface" > gpurun_out/r02b/pytest.log 2>&1
grep -h "rel-L2\|curve\|passed\|failed" gpurun_out/r02b/pytest.log | tail -30
for mode in split fp16; do
EW_BENCH_FULL_BREAKDOWN=1 EW_RESIDUAL=$mode python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline > gpurun_out/r02b/bench_$mode.log 2>&1
grep "ms  avg\|sum of" gpurun_out/r02b/bench_$mode.log | head -12
grep -o '"unet_forward_ms": [0-9.]*' gpurun_out/r02b/bench_$mode.log
done
