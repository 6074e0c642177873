cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4m
python tools/experiments/dbg_ff_folded.py 2>&1 | tail -6
bash tools/experiments/r4m_call.sh
