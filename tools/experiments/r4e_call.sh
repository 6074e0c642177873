cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4e
EW_SKIP_FULL_PARITY=1 timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^tap\|^full-size tap" > gpurun_out/r4e/pytest.log; tail -5 gpurun_out/r4e/pytest.log
grep -n "reference run\|fp32 checkpoint\|navigator \[" gpurun_out/r4e/pytest.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err; tail -c 600 gpurun_out/r4e/bench.json
