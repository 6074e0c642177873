cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4i
run() { # name, env...
  n=$1; shift
  env "$@" EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 3 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4i/breakdown_$n.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n forward ms', d['config']['unet_forward_ms'])"
}
run base A=1
run minm256 EW_G3_MINM=256
run gnb1280 EW_GN_BLOCKS=1280
run gnb2560 EW_GN_BLOCKS=2560
run gnrows16 EW_GN_APPLY_ROWS=16
run gnrows64 EW_GN_APPLY_ROWS=64
run base2 A=1
run minm256b EW_G3_MINM=256
