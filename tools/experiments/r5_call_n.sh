cd $GRAFT_REPO_ROOT; O=gpurun_out/r5n; mkdir -p $O
EW_G3_SK_ALL=1 timeout 1200 python -m pytest tests/test_gpu_fullshape_parity.py tests/test_gpu_gemm_gen3.py -x -q 2>&1 | tail -3
tools/ab_env.sh "EW_G3_SK_ALL=0" "EW_G3_SK_ALL=1" > $O/ab_sk_all.txt 2>&1; cat $O/ab_sk_all.txt
