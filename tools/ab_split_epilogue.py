"""What the split residual stream costs in the GEMM epilogue, per shape: (a) plain fp16 residual and output, (b) split residual in,
plain out, (c) split in and out (the product's default).  python tools/ab_split_epilogue.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from evoworld_amd import ops, _lib
lib = _lib.load()
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K) in ((115200, 640, 640), (115200, 640, 2560), (28800, 1280, 1280), (28800, 1280, 5120), (460800, 320, 320), (460800, 320, 1280)):
    x = torch.rand(M, K, device="cuda", dtype=torch.float16) * 2 - 1
    w = (torch.rand(N, K, device="cuda", dtype=torch.float16) * 2 - 1) * 0.05
    b = torch.rand(N, device="cuda", dtype=torch.float16)
    r = torch.rand(M, N, device="cuda") * 2 - 1
    rs, rp = ops.Res.from_float(r), r.half()
    op, os_ = torch.empty(M, N, dtype=torch.float16, device="cuda"), ops.Res.empty(M, N, "cuda", True)
    ta = timeit(lambda: ops.gemm(x, w, op, M=M, N=N, c1=K, lda=K, bias=b, r1=rp, ld_r1=N)); ka = lib.ew_gemm_last_kernel().decode()
    tb = timeit(lambda: ops.gemm(x, w, op, M=M, N=N, c1=K, lda=K, bias=b, r1=rs, ld_r1=N)); kb = lib.ew_gemm_last_kernel().decode()
    tc = timeit(lambda: ops.gemm(x, w, os_, M=M, N=N, c1=K, lda=K, bias=b, r1=rs, ld_r1=N)); kc = lib.ew_gemm_last_kernel().decode()
    print(f"M={M} N={N} K={K}: plain {ta:7.1f} us [{ka}]  split-in {tb:7.1f} us  split-in/out {tc:7.1f} us [{kc}]")
