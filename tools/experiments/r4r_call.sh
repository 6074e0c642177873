cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4r
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^tap\|^full-size tap" > gpurun_out/r4r/pytest_full.log; tail -4 gpurun_out/r4r/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench forward ms', d['config']['unet_forward_ms'], 'frac', d['roofline']['frac'], 'fps', d['value'])"
