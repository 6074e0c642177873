"""Round 3 A/B: generation 3 with / without the L2 prefetch of a work item's RESIDUAL block during its first K-tile
(ew_set_gemm_debug bit 3 switches it off), same process, interleaved, on the residual-carrying problems of the forward.
Also checks that the results are bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib, ops  # noqa: E402
import tools.bench_kernels as B  # noqa: E402

lib = _lib.load()
EXTRA = int(os.environ.get("EW_DBG_EXTRA", "0"))


def ab(label, flops, fn, out):
    res = {}
    for rnd in range(3):
        for mode in (0, 8):
            lib.ew_set_gemm_debug(mode | EXTRA)
            ms = B.timeit(fn, iters=6, warm=2)
            res.setdefault(mode, []).append(ms)
            if rnd == 0:
                res[("out", mode)] = out.clone() if isinstance(out, torch.Tensor) else out.hi.clone()
    lib.ew_set_gemm_debug(0)
    on, off = min(res[0]), min(res[8])
    same = torch.equal(res[("out", 0)], res[("out", 8)])
    print(f"{label:46s} prefetch {on:7.3f} ms ({flops / on / 1e9:6.0f} TF/s)  off {off:7.3f} ms ({flops / off / 1e9:6.0f} TF/s)  {100 * (off / on - 1):+5.1f} %  "
          f"kernel {lib.ew_gemm_last_kernel().decode()}  identical={same}", flush=True)


def gemm(label, M, N, K, res=False, split=False):
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    out = ops.Res.empty(M, N, "cuda", True) if split else torch.empty(M, N, dtype=torch.float16, device="cuda")
    r1 = (ops.Res.from_float(B.rnd(M, N).float()) if split else B.rnd(M, N)) if res else None
    ab(f"dense {label} {M}x{N}x{K}", 2.0 * M * N * K, lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, r1=r1, ld_r1=N if res else 0), out)


def conv(label, N, C, O, H, W, c2=0, split=False, up=0, stride=1):
    x = B.rnd(N * H * W, C)
    x2 = B.rnd(N * H * W, c2) if c2 else None
    Ho, Wo = (H // 2, W // 2) if stride == 2 else ((2 * H, 2 * W) if up else (H, W))
    M = N * Ho * Wo
    w, b = B.rnd(O, 9 * (C + c2)) * 0.02, B.rnd(O)
    out = ops.Res.empty(M, O, "cuda", True) if split else torch.empty(M, O, dtype=torch.float16, device="cuda")
    r1 = ops.Res.from_float(B.rnd(M, O).float()) if split else None
    kw = dict(r1=r1, ld_r1=O) if split else {}
    ab(f"conv3x3 {label} {C}+{c2}->{O} @{H}x{W}", 2.0 * M * O * 9 * (C + c2),
       lambda: ops.gemm(x, w, out, M=M, N=O, c1=C, lda=C, a2=x2, c2=c2, lda2=c2, bias=b, mode=ops.A_CONV3X3, conv=(N, H, W, Ho, Wo, stride, up), **kw), out)


def convt(label, Bn, T, P, C, split=False):
    M = Bn * T * P
    x, w, b = B.rnd(M, C), B.rnd(C, 3 * C) * 0.02, B.rnd(C)
    out = ops.Res.empty(M, C, "cuda", True) if split else torch.empty(M, C, dtype=torch.float16, device="cuda")
    r1 = ops.Res.from_float(B.rnd(M, C).float()) if split else None
    kw = dict(r1=r1, ld_r1=C) if split else {}
    ab(f"convT3 {label} C={C} P={P}", 2.0 * M * C * 3 * C, lambda: ops.gemm(x, w, out, M=M, N=C, c1=C, lda=C, bias=b, mode=ops.A_CONVT3, tconv=(Bn, T, P), **kw), out)




def gemm2r(label, M, N, K):
    """dense + row-bias + two split residuals (<0,23>: the AlphaBlender epilogue of the temporal feed-forward)"""
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    r1, r2 = B.rnd(M, N), ops.Res.from_float(B.rnd(M, N).float())
    ab(f"dense 2res {label} {M}x{N}x{K}", 2.0 * M * N * K,
       lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, r1=r1, ld_r1=N, r2=r2, ld_r2=N, c_acc=0.5, c_r1=0.5, c_r2=0.5), out)


gemm("L1 proj", 115200, 640, 640, res=True, split=True)
gemm("L1 ff_down", 115200, 640, 2560, res=True, split=True)
gemm("L2 proj", 28800, 1280, 1280, res=True, split=True)
gemm("L2 ff_down", 28800, 1280, 5120, res=True, split=True)
gemm("L0 ff_down EW_G3_SHORT", 460800, 320, 1280, res=True, split=True)
gemm2r("L1 t_ff_down", 115200, 640, 2560)
gemm2r("L2 t_ff_down", 28800, 1280, 5120)
conv("L0 split res", 50, 320, 320, 72, 128, split=True)
conv("L0 cat960 split", 50, 640, 320, 72, 128, c2=320, split=True)
conv("L1 split res", 50, 640, 640, 36, 64, split=True)
conv("L2 split res", 50, 1280, 1280, 18, 32, split=True)
convt("L0 split", 2, 25, 9216, 320, split=True)
convt("L1 split", 2, 25, 2304, 640, split=True)
convt("L2 split", 2, 25, 576, 1280, split=True)
