#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_gen3.py -q -m gpu -x -s -k streamk > $O/pytest_sk.log 2>&1
grep -h "stream-K\|passed\|failed\|Error" $O/pytest_sk.log | tail -12
timeout 900 python -m pytest tests/test_gpu_gemm_gen3.py tests/test_gpu_ops.py tests/test_gpu_unet.py -q -m gpu -x > $O/pytest_a.log 2>&1
tail -3 $O/pytest_a.log
for s in 0 1; do
EW_G3_SK=$s EW_BENCH_FULL_BREAKDOWN=1 timeout 900 python bench.py --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline > $O/bench$s.log 2>&1
grep -v '^{' $O/bench$s.log | tail -27
grep '^{' $O/bench$s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('EW_G3_SK=$s forward ms', d['config']['unet_forward_ms'])"
done
python -c "
from evoworld_amd import _lib
print('streamk status', _lib.load().ew_gemm_streamk_status())"
