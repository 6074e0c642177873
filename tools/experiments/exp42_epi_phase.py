"""Round 5: where the time of the residual-carrying generation-3 kernels goes -- full kernel vs stores skipped (debug bit 1) vs epilogue
skipped (bit 2), on the split-stream shapes of the U-Net.  Run once per library: EW_LIB_PATH=... python tools/experiments/exp42_epi_phase.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from evoworld_amd import _lib, ops
lib = _lib.load()
DEV = "cuda"
g = torch.Generator(device="cpu").manual_seed(3)
rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)


def timeit(fn, iters=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def dense(M, N, K, rb=False):
    x, w, b = rnd(M, K).half().to(DEV), (rnd(N, K) / K ** 0.5).half().to(DEV), rnd(N).half().to(DEV)
    r1 = ops.Res.from_float(rnd(M, N).to(DEV) * 3)
    out = ops.Res.empty(M, N, DEV, True)
    kw = dict(rowbias=rnd(2, N).half().to(DEV), rows_per_group=M // 2, ld_rowbias=N) if rb else {}
    return lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, r1=r1, ld_r1=N, **kw)


def conv(n, C, H, W, res=True):
    M = n * H * W
    x, w, b = rnd(M, C).half().to(DEV), (rnd(C, 9 * C) / (9 * C) ** 0.5).half().to(DEV), rnd(C).half().to(DEV)
    r1 = ops.Res.from_float(rnd(M, C).to(DEV) * 3) if res else None
    out = ops.Res.empty(M, C, DEV, True)
    kw = {} if res else dict(rowbias=rnd(n, C).half().to(DEV), rows_per_group=H * W, ld_rowbias=C)
    return lambda: ops.gemm(x, w, out, M=M, N=C, c1=C, lda=C, bias=b, mode=ops.A_CONV3X3, conv=(n, H, W, H, W, 1, 0), r1=r1, ld_r1=C if res else 0, **kw)


def convt(B, T, P, C):
    M = B * T * P
    x, w, b = rnd(M, C).half().to(DEV), (rnd(C, 3 * C) / (3 * C) ** 0.5).half().to(DEV), rnd(C).half().to(DEV)
    r1 = ops.Res.from_float(rnd(M, C).to(DEV) * 3)
    out = ops.Res.empty(M, C, DEV, True)
    return lambda: ops.gemm(x, w, out, M=M, N=C, c1=C, lda=C, bias=b, mode=ops.A_CONVT3, tconv=(B, T, P), r1=r1, ld_r1=C)


def main():
    cases = [("dense 115200x640x640 rb", lambda: dense(115200, 640, 640, True)), ("dense 28800x1280x1280 rb", lambda: dense(28800, 1280, 1280, True)),
             ("dense 115200x640x2560", lambda: dense(115200, 640, 2560)), ("dense 28800x1280x5120", lambda: dense(28800, 1280, 5120)),
             ("convT 460800x320x960", lambda: convt(2, 25, 9216, 320)), ("conv3x3 L0 460800x320x2880 res", lambda: conv(50, 320, 72, 128)),
             ("conv3x3 L1 115200x640x5760 res", lambda: conv(50, 640, 36, 64)), ("conv3x3 L0 460800x320x2880 rb", lambda: conv(50, 320, 72, 128, False))]
    for name, mk in cases:
        fn = mk()
        row = []
        for dbg in (0, 1, 2, 0, 1, 2):
            lib.ew_set_gemm_debug(dbg)
            row.append(timeit(fn))
        lib.ew_set_gemm_debug(0)
        fn()
        k = lib.ew_gemm_last_kernel().decode()
        print(f"{name:36s} {k:22s} full {min(row[0], row[3]):7.1f}  no-stores {min(row[1], row[4]):7.1f}  no-epilogue {min(row[2], row[5]):7.1f} us", flush=True)
        del fn
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
