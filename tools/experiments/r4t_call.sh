cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4t
R=$GRAFT_REPO_ROOT
for L in "" $R/evoworld_amd/libevoworld_hip_ap1.so $R/evoworld_amd/libevoworld_hip_ap2.so "" $R/evoworld_amd/libevoworld_hip_ap1.so $R/evoworld_amd/libevoworld_hip_ap2.so; do
echo "lib=$(basename ${L:-base})"; EW_LIB_PATH=$L ITERS=5 REPS=1 python tools/attn_bench.py 2>/dev/null | grep log2
done > gpurun_out/r4t/attn_prio.txt; cat gpurun_out/r4t/attn_prio.txt
