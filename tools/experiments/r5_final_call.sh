cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
EW_FULL_PARITY_STEPS=25 EW_FULL_FP32_WEIGHTS=1 EW_FULL_PARITY_CKPT=tests/_ckpt/clip_oracle_fp32w.pt timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -s -k full_size_clip > gpurun_out/r05_k_full_size_clip.log 2>&1; grep -E "rel-L2|passed|failed|Error" gpurun_out/r05_k_full_size_clip.log | tail -14
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -k "tiny or split" 2>&1 | tail -2
bash tools/run_record.sh > gpurun_out/r05_j_record.log 2>&1; tail -30 gpurun_out/r05_j_record.log
