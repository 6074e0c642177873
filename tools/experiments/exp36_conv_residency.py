"""Round 3: what bounds the 3x3 convs (1050-1180 TF/s against ~1400 for MFMA + ds_read alone)?  Full / no-DMA / DMA-only builds
with the real input and with lda = 0 (every pixel aliases pixel 0: the A stream is one cache-resident row; W unchanged)."""
import ctypes
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


def case(N, C, O, H, W, lda):
    x = B.rnd(N * H * W, C)
    w, b = B.rnd(O, 9 * C) * 0.02, B.rnd(O)
    M = N * H * W
    out = torch.empty(M, O, dtype=torch.float16, device="cuda")
    row = []
    for name, fn in libs:
        new.ew_gemm_f16 = fn
        row.append(B.timeit(lambda: ops.gemm(x, w, out, M=M, N=O, c1=C, lda=lda, bias=b, mode=ops.A_CONV3X3, conv=(N, H, W, H, W, 1, 0)), iters=6, warm=2))
    new.ew_gemm_f16 = prod
    fl = 2.0 * M * O * 9 * C
    print(f"conv {C}->{O} @{H}x{W} lda={lda:4d}: full {row[0]:7.3f} ms ({fl / row[0] / 1e9:6.0f} TF/s)  noDMA {row[1]:7.3f} ({fl / row[1] / 1e9:6.0f})  DMAonly {row[2]:7.3f} ms", flush=True)


for rnd in range(2):
    for (N, C, O, H, W) in ((50, 320, 320, 72, 128), (50, 640, 640, 36, 64), (50, 1280, 1280, 18, 32)):
        case(N, C, O, H, W, C)
        case(N, C, O, H, W, 0)
