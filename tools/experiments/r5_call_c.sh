# round 5, call C: hygiene changes on the GPU (stream-K naming, half-split dispatch, tightened tolerances, 4-rank CFG test), smoke, effective clock per kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c
timeout 1500 python -m pytest tests/test_gpu_gemm_gen3.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_unet.py tests/test_gpu_vae.py -m gpu -q -x -k "not full_size" 2>&1 | tail -5
timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -2
bash tools/pmc_clock.sh r05_c 2>&1 | tee gpurun_out/r5c/clock.txt
timeout 900 python tools/experiments/exp43_stagger.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c/exp43_stagger.txt
