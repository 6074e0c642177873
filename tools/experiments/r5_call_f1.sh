# round 5, call F1: full-size 25-step clip under the SURVEY 8d weight protocol (oracle on un-rounded fp32 weights), oracle steps 1..15
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5f
EW_FULL_FP32_WEIGHTS=1 EW_FULL_PARITY_STEPS=25 EW_FULL_PARITY_STOP=15 EW_ORACLE_THREADS=48 timeout 3500 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s -k full_size_clip 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f/fullsize_fp32_clip_part1.log | tail -25
