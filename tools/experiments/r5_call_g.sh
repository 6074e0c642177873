cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "split or denormal or layout or euler or dual" -s > gpurun_out/r5g/ops.log 2>&1; tail -5 gpurun_out/r5g/ops.log
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -s -k "not full_size" > gpurun_out/r5g/unet.log 2>&1; grep -E "rel-L2|passed|failed|Error|error" gpurun_out/r5g/unet.log | tail -40
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_pipeline_glue.py -x -q -s > gpurun_out/r5g/pipe.log 2>&1; grep -E "rel-L2|passed|failed|Error|error|curve" gpurun_out/r5g/pipe.log | tail -30
tools/ab_env.sh "EW_SPLIT_OPERANDS=0" "EW_SPLIT_OPERANDS=1" > gpurun_out/r5g/ab_split.txt 2>&1; cat gpurun_out/r5g/ab_split.txt
