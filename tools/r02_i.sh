#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm_gen3.py tests/test_gpu_fullshape_parity.py tests/test_gpu_unet.py tests/test_gpu_clip.py -q -m gpu -x > $O/pytest_a.log 2>&1
tail -3 $O/pytest_a.log
cd $R/_old_r01g && python $R/tools/ab_kernels.py 2>&1 | tail -18 > $O/old.log
cd $R && python tools/ab_kernels.py 2>&1 | tail -18 > $O/new.log
paste $O/old.log $O/new.log | cut -c1-160
for s in 0 1; do
EW_G3_SHORT=$s EW_BENCH_FULL_BREAKDOWN=1 python bench.py --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline > $O/bench$s.log 2>&1
grep -v '^{' $O/bench$s.log | tail -28
grep '^{' $O/bench$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('EW_G3_SHORT=$s forward ms', d['config']['unet_forward_ms'])"
done
