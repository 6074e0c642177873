cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4u
run() { n=$1; shift
  env "$@" EW_BENCH_BY_SHAPE=1 EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline --no-fp16-stream 2> gpurun_out/r4u/bd_$n.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n forward ms', d['config']['unet_forward_ms'])"
}
run base A=1
run short1 EW_G3_SHORT=1
run sk2 EW_G3_SK=2
run base2 A=1
run short1b EW_G3_SHORT=1
run skmink640 EW_G3_SK_MINK=640
grep -h "M=460800 N=320 K=320 \|M=460800 N=320 K=640 " gpurun_out/r4u/bd_base.txt gpurun_out/r4u/bd_short1.txt
