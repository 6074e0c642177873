# Box classifier in front of tools/run_record.sh: gpurun boxes differ by 3-5 %; the level-0 spatial attention launch (35 ms of the forward)
# is measured first and the full record is taken only when the box is of the faster class (< $1 TF/s threshold given as argument).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
tf=$(python tools/attn_bench.py 2>/dev/null | tail -1 | awk '{print $(NF-1)}')
echo "box class probe: attn_spatial $tf TF/s (threshold $1)" | tee gpurun_out/r03_boxprobe.txt
if python -c "import sys; sys.exit(0 if float('$tf') >= float('$1') else 1)"; then bash tools/run_record.sh; else echo "slow box: record skipped"; fi
