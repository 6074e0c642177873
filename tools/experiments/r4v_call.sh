cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4v
timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py tests/test_gpu_unet.py -m gpu -q -x -k "not full_size" 2>&1 | tail -3
run() { n=$1; shift
  env "$@" EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4v/bd_$n.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n forward ms', d['config']['unet_forward_ms'])"
}
run perm1 EW_G3_TPERM=1
run perm0 EW_G3_TPERM=0
run perm1b EW_G3_TPERM=1
run perm0b EW_G3_TPERM=0
for n in perm0 perm1; do echo == $n; grep -E "gemm3_kernel<2," gpurun_out/r4v/bd_$n.txt; done
