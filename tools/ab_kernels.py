#!/usr/bin/env python3
"""A/B micro-benchmark that runs unchanged in this tree and in an older checkout (only arguments ew_gemm_f16 has had since
round 1): python tools/ab_kernels.py  (run from the tree root to be measured)."""
import os
import sys
import torch

sys.path.insert(0, os.getcwd())
from evoworld_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*shape):
    return torch.rand(*shape, device=DEV, dtype=torch.float16) * 2 - 1


def conv(name, n, C, O, H, W, rowbias=False, res=False):
    x, w, b = rnd(n * H * W, C), rnd(O, 9 * C) * 0.02, rnd(O)
    out = torch.empty(n * H * W, O, dtype=torch.float16, device=DEV)
    rb = rnd(n, O) if rowbias else None
    r1 = rnd(n * H * W, O) if res else None
    kw = dict(rowbias=rb, rows_per_group=H * W, ld_rowbias=O) if rowbias else {}
    fn = lambda: ops.gemm(x, w, out, M=n * H * W, N=O, c1=C, lda=C, bias=b, mode=ops.A_CONV3X3, conv=(n, H, W, H, W, 1, 0),
                          r1=r1, ld_r1=O if res else 0, **kw)
    ms = timeit(fn)
    print(f"conv {name:24s} {ms * 1e3:8.1f} us  {2.0 * n * H * W * O * 9 * C / ms / 1e9:7.1f} TF/s", flush=True)


def gemm(name, M, N, K, act=0, res=False, rowbias=False):
    x, w, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
    out = torch.empty(M, N // 2 if act == 2 else N, dtype=torch.float16, device=DEV)
    r1 = rnd(M, N) if res else None
    kw = dict(rowbias=rnd(2, N), rows_per_group=M // 2, ld_rowbias=N) if rowbias else {}
    fn = lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, act=act, r1=r1, ld_r1=N if res else 0, **kw)
    ms = timeit(fn)
    print(f"gemm {name:24s} {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)


for rep in range(2):
    conv("L0 320 rowbias", 50, 320, 320, 72, 128, rowbias=True)
    conv("L1 640 rowbias", 50, 640, 640, 36, 64, rowbias=True)
    conv("L2 1280 rowbias", 50, 1280, 1280, 18, 32, rowbias=True)
    conv("L0 320 res", 50, 320, 320, 72, 128, res=True)
    conv("L1 640 res", 50, 640, 640, 36, 64, res=True)
    conv("L0 320 plain", 50, 320, 320, 72, 128)
    for lvl, (tok, C) in enumerate(((460800, 320), (115200, 640), (28800, 1280))):
        gemm(f"L{lvl} geglu", tok, 8 * C, C, act=2)
        gemm(f"L{lvl} ff_down res", tok, C, 4 * C, res=True)
        gemm(f"L{lvl} proj res rowbias", tok, C, C, res=True, rowbias=True)
        gemm(f"L{lvl} qkv", tok, 3 * C, C)
