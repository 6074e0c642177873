cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4l
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^tap\|^full-size tap" > gpurun_out/r4l/pytest_full.log; tail -4 gpurun_out/r4l/pytest_full.log
EW_FULL_FP32_WEIGHTS=1 timeout 1200 python -m pytest tests/test_gpu_unet.py -m gpu -q -s -k "full_size_forward" 2>&1 | grep -v "^tap" > gpurun_out/r4l/fullsize_fp32_weights.log; tail -12 gpurun_out/r4l/fullsize_fp32_weights.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
