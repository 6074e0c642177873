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
for (M, N, K) in ((115200, 640, 640), (28800, 1280, 1280)):
    x = torch.rand(M, K, device="cuda", dtype=torch.float16) * 2 - 1
    w = (torch.rand(N, K, device="cuda", dtype=torch.float16) * 2 - 1) * 0.05
    b = torch.rand(N, device="cuda", dtype=torch.float16)
    r = torch.rand(M, N, device="cuda") * 2 - 1
    rs, rp = ops.Res.from_float(r), r.half()
    op, os_ = torch.empty(M, N, dtype=torch.float16, device="cuda"), ops.Res.empty(M, N, "cuda", True)
    t0 = timeit(lambda: ops.gemm(x, w, op, M=M, N=N, c1=K, lda=K, bias=b))
    ta = timeit(lambda: ops.gemm(x, w, op, M=M, N=N, c1=K, lda=K, bias=b, r1=rp, ld_r1=N))
    tab = timeit(lambda: ops.gemm(x, w, op, M=M, N=N, c1=K, lda=K, bias=b, r1=rp, ld_r1=0))
    tb = timeit(lambda: ops.gemm(x, w, op, M=M, N=N, c1=K, lda=K, bias=b, r1=rs, ld_r1=N))
    tbb = timeit(lambda: ops.gemm(x, w, op, M=M, N=N, c1=K, lda=K, bias=b, r1=rs, ld_r1=0))
    tc = timeit(lambda: ops.gemm(x, w, os_, M=M, N=N, c1=K, lda=K, bias=b, r1=rs, ld_r1=N))
    tcb = timeit(lambda: ops.gemm(x, w, os_, M=M, N=N, c1=K, lda=K, bias=b, r1=rs, ld_r1=0))
    tco = timeit(lambda: ops.gemm(x, w, os_, M=M, N=N, c1=K, lda=K, bias=b))
    print(f"M={M} N={N} K={K}: no-res {t0:.1f} | plain r1 {ta:.1f} (r1 from one L2 row {tab:.1f}) | split-in {tb:.1f} ({tbb:.1f}) | split-in/out {tc:.1f} ({tcb:.1f}) | split-out only {tco:.1f}")
