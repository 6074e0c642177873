# round 5, call F2: ... oracle steps 16..25, resumed from the latents checkpoint of call F1 (copied into the tree: gpurun_out/ does not travel)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5f
EW_FULL_FP32_WEIGHTS=1 EW_FULL_PARITY_STEPS=25 EW_FULL_PARITY_CKPT=tests/_ckpt/clip_oracle_ckpt.pt EW_ORACLE_THREADS=48 timeout 3500 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s -k full_size_clip 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f/fullsize_fp32_clip_part2.log | tail -25
