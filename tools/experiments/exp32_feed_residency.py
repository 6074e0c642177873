"""Round 3: is generation 3's LDS-DMA feed (~54 GB/s per CU in the DMA-only ablation build) bound by the DMA path or by where the
operand lines come from?  tools/experiments/mb_feed.hip measures 108 (one 72 KB tile in flight) ... 131 GB/s per CU (two) for a
2 MB L2-resident region per XCD.  Here the DMA-only / no-DMA / full builds run real shapes with (a) the real A operand and
(b) lda = 0 (every A row aliases row 0: the A stream is 2K bytes, always cache-resident) -- same instruction stream, different
memory side."""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib, ops  # noqa: E402
import tools.bench_kernels as B  # noqa: E402

new = _lib.load()
prod = new.ew_gemm_f16
d = os.path.dirname(_lib.__file__)
libs = [("full", prod)]
for a, what in ((4, "no DMA"), (3, "DMA only")):
    L = ctypes.CDLL(os.path.join(d, f"libevoworld_hip_g3ab{a}.so"))
    L.ew_gemm_f16.argtypes = prod.argtypes
    L.ew_gemm_f16.restype = prod.restype
    libs.append((what, L.ew_gemm_f16))


def case(M, N, K, lda, act=0):
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    out = torch.empty(M, N // 2 if act == 2 else N, dtype=torch.float16, device="cuda")
    row = []
    for name, fn in libs:
        new.ew_gemm_f16 = fn
        ms = B.timeit(lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=lda, bias=b, act=act), iters=8, warm=3)
        row.append(ms)
    new.ew_gemm_f16 = prod
    tiles = math.ceil(M / 256) * (N // 320)
    staged = tiles * (K // 64) * 576 * 128
    print(f"M={M:6d} N={N:5d} K={K:4d} lda={lda:4d}: full {row[0]:7.3f} ms ({2.0 * M * N * K / row[0] / 1e9:6.0f} TF/s)  noDMA {row[1]:7.3f}  "
          f"DMAonly {row[2]:7.3f} ms = {staged / row[2] / 1e6 / 256:6.1f} GB/s per CU staged ({staged / row[2] / 1e9:5.1f} TB/s chip), "
          f"W {N * K * 2 / 1e6:5.1f} MB  A {M * K * 2 / 1e6 if lda else 0:6.1f} MB", flush=True)


for rnd in range(2):
    for (M, N, K, act) in ((28800, 1280, 5120, 0), (65536, 320, 5120, 0), (115200, 5120, 640, 2), (460800, 2560, 320, 2),
                           (115200, 640, 2560, 0), (65536, 320, 1280, 0)):
        case(M, N, K, K, act)
        case(M, N, K, 0, act)
