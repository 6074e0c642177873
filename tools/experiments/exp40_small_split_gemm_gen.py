"""Round 3: generation 2 vs 3 on the small residual-carrying dense GEMMs of levels 1-2 (split stream in and out), one process per
generation (EW_GEMM_GEN is read once).  Usage: EW_GEMM_GEN=2|3 python tools/experiments/exp40_small_split_gemm_gen.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import _lib, ops  # noqa: E402
import tools.bench_kernels as B  # noqa: E402

lib = _lib.load()
for M, N, K in ((115200, 640, 640), (28800, 1280, 1280), (115200, 640, 2560), (28800, 1280, 5120), (460800, 320, 320), (460800, 320, 640)):
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    out = ops.Res.empty(M, N, "cuda", True)
    r1 = ops.Res.from_float(B.rnd(M, N).float())
    fn = lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, r1=r1, ld_r1=N)
    ms = min(B.timeit(fn, iters=8, warm=3) for _ in range(3))
    print(f"gen {os.environ.get('EW_GEMM_GEN', 'auto'):4s} {M}x{N}x{K}: {ms * 1e3:7.1f} us  {2.0 * M * N * K / ms / 1e9:6.0f} TF/s  {(M * K * 2 + M * N * 6) / ms / 1e6:6.0f} GB/s  {lib.ew_gemm_last_kernel().decode()}", flush=True)
