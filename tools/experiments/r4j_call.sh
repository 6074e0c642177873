cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4j
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gemm_gen3.py -m gpu -q -x -k "dense_epilogues" 2>&1 | tail -2
run() { n=$1; shift
  env "$@" EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 3 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4j/breakdown_$n.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n forward ms', d['config']['unet_forward_ms'])"
}
run base A=1
run u8 EW_LIB_PATH=$R/evoworld_amd/libevoworld_hip_gnu8.so
run rows16 EW_GN_APPLY_ROWS=16
run rows8 EW_GN_APPLY_ROWS=8
run u8rows16 EW_LIB_PATH=$R/evoworld_amd/libevoworld_hip_gnu8.so EW_GN_APPLY_ROWS=16
run base2 A=1
for f in base u8 rows16 rows8 u8rows16 base2; do echo == $f; grep -E "gn_(apply|stats)_kernel" gpurun_out/r4j/breakdown_$f.txt | sort -k9 -rn | head -6; done
