cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4x
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^tap\|^full-size tap" > gpurun_out/r4x/pytest_full.log; tail -3 gpurun_out/r4x/pytest_full.log
bash tools/run_record.sh > gpurun_out/r4x/record.log 2>&1; tail -c 400 gpurun_out/r04_h_bench.json
