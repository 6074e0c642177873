cd $GRAFT_REPO_ROOT; O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py -x -q -k "conv3x3" > $O/halo_test.log 2>&1; tail -15 $O/halo_test.log
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -k "tiny_vs_oracle or bit_repro" 2>&1 | tail -3
tools/ab_env.sh "EW_G3_HALO=0" "EW_G3_HALO=1" > $O/ab_halo.txt 2>&1; cat $O/ab_halo.txt
