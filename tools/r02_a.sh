#!/bin/bash
mkdir -p gpurun_out/r02a
for mode in split split_min fp16; do
EW_RESIDUAL=$mode python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_pipeline.py -q -m gpu -s > gpurun_out/r02a/pytest_$mode.log 2>&1
grep -h "rel-L2\|curve\|passed\|failed" gpurun_out/r02a/pytest_$mode.log | tail -30
done
python -m pytest tests/test_gpu_gemm_gen3.py tests/test_gpu_fullshape_parity.py -q -m gpu -s > gpurun_out/r02a/pytest_b.log 2>&1
grep -h "rel-L2\|curve\|passed\|failed" gpurun_out/r02a/pytest_b.log | tail -30
EW_RESIDUAL=split_min python bench.py --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline > gpurun_out/r02a/bench_split_min.log 2>&1
tail -1 gpurun_out/r02a/bench_split_min.log | cut -c1-3000
