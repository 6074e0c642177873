import os, sys, torch
sys.path.insert(0, os.getcwd())
from evoworld_amd import ops
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for rep in range(2):
    for (M, N, K) in ((460800, 2560, 320), (115200, 5120, 640), (28800, 10240, 1280)):
        x = torch.rand(M, K, device="cuda", dtype=torch.float16) * 2 - 1
        w = (torch.rand(N, K, device="cuda", dtype=torch.float16) * 2 - 1) * 0.05
        b = torch.rand(N, device="cuda", dtype=torch.float16)
        out = torch.empty(M, N // 2, dtype=torch.float16, device="cuda")
        ms = timeit(lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, act=2))
        print(f"geglu M={M} N={N} K={K}: {ms*1e3:.1f} us")
