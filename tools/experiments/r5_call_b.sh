cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py -m gpu -q -x 2>&1 | tail -3
for l in base "" base ""; do
echo "## lib=${l:-new}"
L=""; [ -n "$l" ] && L=$GRAFT_REPO_ROOT/evoworld_amd/libevoworld_hip_$l.so
EW_LIB_PATH=$L timeout 600 python tools/experiments/exp42_epi_phase.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5b/exp42.txt
