cd $GRAFT_REPO_ROOT; O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullshape_parity.py -x -q -k "attn" 2>&1 | tail -3
for r in 1 2; do for L in attn_late ""; do P=""; [ -n "$L" ] && P=$PWD/evoworld_amd/libevoworld_hip_$L.so; echo "## lib=${L:-new(early write)}"; EW_LIB_PATH=$P REPS=1 timeout 300 python tools/attn_bench.py 2>&1 | grep TF; for s in 2304 576; do EW_LIB_PATH=$P REPS=1 S=$s timeout 300 python tools/attn_bench.py 2>&1 | grep log2.*TF; done; done; done | tee $O/attn_early_write_bench.txt
EW_LIB_PATH=$PWD/evoworld_amd/libevoworld_hip_attn_trace.so timeout 600 python tools/experiments/exp47_attn_trace.py > $O/exp47_attn_trace_early.txt 2>&1; cat $O/exp47_attn_trace_early.txt
tools/ab_lib.sh evoworld_amd/libevoworld_hip_attn_late.so > $O/ab_attn_early.txt 2>&1; cat $O/ab_attn_early.txt | head -30
