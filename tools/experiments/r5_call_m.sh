cd $GRAFT_REPO_ROOT; O=gpurun_out/r5m; mkdir -p $O
timeout 3000 python -m pytest tests/ -q -m gpu -x > $O/gpu_suite_full.log 2>&1; tail -5 $O/gpu_suite_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
