#!/bin/bash
python -m pytest tests/test_gpu_gemm_gen3.py -q -m gpu -x -k "short_k" 2>&1 | grep -v amdgpu | tail -12
