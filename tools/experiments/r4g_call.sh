cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4g
R=$GRAFT_REPO_ROOT
for L in "" $R/evoworld_amd/libevoworld_hip_rs1.so "" $R/evoworld_amd/libevoworld_hip_rs1.so; do
echo "lib=${L:-default}"; EW_LIB_PATH=$L ITERS=5 REPS=1 python tools/attn_bench.py
done > gpurun_out/r4g/attn_ab.txt 2>&1
cat gpurun_out/r4g/attn_ab.txt
for L in "" $R/evoworld_amd/libevoworld_hip_rs1.so; do
echo "lib=${L:-default}"; EW_LIB_PATH=$L timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullshape_parity.py -m gpu -q -s -k "attn_spatial" 2>&1 | grep -i "rel-L2\|passed\|failed"
done > gpurun_out/r4g/attn_parity.txt 2>&1
cat gpurun_out/r4g/attn_parity.txt
for L in "" $R/evoworld_amd/libevoworld_hip_rs1.so; do
echo "lib=${L:-default}"; EW_LIB_PATH=$L ITERS=1 REPS=1 NSEQ=10 bash tools/pmc_run.sh $R/tools/attn_bench.py attn_spatial
done > gpurun_out/r4g/attn_pmc.txt 2>&1
cat gpurun_out/r4g/attn_pmc.txt
for s in new old new old; do
L=""; [ $s = new ] && L=$R/evoworld_amd/libevoworld_hip_rs1.so
EW_LIB_PATH=$L timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s (rs1=new) forward ms', d['config']['unet_forward_ms'])"
done > gpurun_out/r4g/fwd_ab.txt
cat gpurun_out/r4g/fwd_ab.txt
timeout 600 python -m pytest tests/test_gpu_pipeline_glue.py tests/test_gpu_pipeline.py tests/test_gpu_gemm_gen3.py -m gpu -q -x -k "batch_of_two or cfg or streamk" 2>&1 | tail -3
