# round 5, call D: spatial attention without the per-tile max (lazy max) + V^T tile by ds_write_b64: tests, stand-alone bench A/B, forward A/B; stagger re-measured with an interleaved protocol; short rule
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullshape_parity.py -m gpu -q -x -k "attn" 2>&1 | tail -12
for l in base "" base ""; do
echo "## lib=${l:-new}"
L=""; [ -n "$l" ] && L=$GRAFT_REPO_ROOT/evoworld_amd/libevoworld_hip_$l.so
EW_LIB_PATH=$L ITERS=5 timeout 600 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5d/attn_bench.txt
bash tools/ab_lib.sh evoworld_amd/libevoworld_hip_base.so 2>&1 | tee gpurun_out/r5d/ab_forward.txt
timeout 900 python tools/experiments/exp43_stagger.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5d/exp43_stagger.txt
for sr in 0 1 0 1; do EW_G3_SHORT=$sr timeout 300 python tools/experiments/exp44_short_rule.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5d/exp44_short.txt
bash tools/pmc_clock.sh r05_d 2>&1 | tee gpurun_out/r5d/clock.txt
