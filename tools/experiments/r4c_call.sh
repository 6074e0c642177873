cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_ops.py -q -k "attn_spatial" -s 2>&1 | tail -15 > gpurun_out/r4c/test_attn.txt
python tools/attn_bench.py > gpurun_out/r4c/attn_bench_b2.txt 2>&1
EW_LIB_PATH=$GRAFT_REPO_ROOT/evoworld_amd/libevoworld_hip_attnb3.so python tools/attn_bench.py > gpurun_out/r4c/attn_bench_b3.txt 2>&1
S=2304 NSEQ=50 python tools/attn_bench.py > gpurun_out/r4c/attn_bench_b2_s2304.txt 2>&1
python -m pytest tests/test_gpu_unet.py -q -s -k "tiny or T25 or t25" 2>&1 | tail -15 > gpurun_out/r4c/test_unet.txt
cat gpurun_out/r4c/*.txt
