cd $GRAFT_REPO_ROOT; O=gpurun_out/r5k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_gen3.py -x -q -k "halo" 2>&1 | tail -2
EW_LIB_PATH=$PWD/evoworld_amd/libevoworld_hip_attn_trace.so timeout 600 python tools/experiments/exp47_attn_trace.py > $O/exp47_attn_trace.txt 2>&1; cat $O/exp47_attn_trace.txt
timeout 300 python tools/attn_bench.py 2>&1 | tail -3
tools/ab_env.sh "EW_G3_SHORT=0" "EW_G3_SHORT=1" > $O/ab_short.txt 2>&1; cat $O/ab_short.txt
