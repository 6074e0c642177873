#!/bin/bash
# lo8 split stream: kernel tests + model parity + quick bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullshape_parity.py tests/test_gpu_vae.py tests/test_gpu_clip.py tests/test_gpu_unet.py tests/test_gpu_gemm_gen3.py -q -m gpu -s -x > $O/pytest_a.log 2>&1
grep -h "rel-L2\|passed\|failed\|Error\|error" $O/pytest_a.log | tail -40
python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s > $O/pytest_b.log 2>&1
grep -h "rel-L2\|curve\|passed\|failed" $O/pytest_b.log | tail -20
EW_BENCH_FULL_BREAKDOWN=1 python bench.py --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline > $O/bench.log 2>&1
grep -v '^{' $O/bench.log | tail -32
grep '^{' $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['unet_forward_ms'], d['roofline']['frac'])"
