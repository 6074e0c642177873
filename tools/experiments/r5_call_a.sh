# round 5, call A: generation-3 RES_LDS epilogue (residual tiles through LDS) + row-bias in the accumulator init: tests, then A/B against the round-4 library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_gemm_gen3.py tests/test_gpu_ops.py tests/test_gpu_fullshape_parity.py tests/test_gpu_ff_fused.py -m gpu -q -x 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -k "not full_size" 2>&1 | tail -5
bash tools/ab_lib.sh evoworld_amd/libevoworld_hip_base.so 2>&1 | tee gpurun_out/r5a/ab.txt
