"""generation 3 vs generation 4 (ew_set_gemm_generation) on the GEGLU up-projection shapes and their bias-only twins: time, TF/s, agreement."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from evoworld_amd import _lib, ops
lib = _lib.load()
g = torch.Generator().manual_seed(0)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (M, N, K, act) in ((28800, 10240, 1280, 2), (115200, 5120, 640, 2), (28800, 10240, 1280, 0), (28800, 1280, 5120, 0), (8192, 8192, 8192, 0), (28800 - 100, 2560, 1280, 0)):
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    n_out = N // 2 if act == 2 else N
    o3 = torch.empty(M, n_out, dtype=torch.float16, device="cuda"); o4 = torch.empty_like(o3)
    res = {}
    for gen, o in ((3, o3), (4, o4)):
        lib.ew_set_gemm_generation(gen)
        o.fill_(7.0)
        ms = t(lambda: ops.gemm(x, w, o, M=M, N=N, c1=K, lda=K, bias=b, act=act))
        res[gen] = (ms, lib.ew_gemm_last_kernel().decode())
    d = float((o3.float() - o4.float()).norm() / o3.float().norm())
    # same launches with every A row aliased onto row 0 (lda = 0: A always an L2 hit; results meaningless): what the A stream's HBM latency costs
    al = {}
    for gen in (3, 4):
        lib.ew_set_gemm_generation(gen)
        al[gen] = t(lambda: ops.gemm(x, w, o4, M=M, N=N, c1=K, lda=0, bias=b, act=act))
    lib.ew_set_gemm_generation(3)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} act={act}: gen3 {res[3][1]} {res[3][0]:.3f} ms {fl / res[3][0] / 1e9:.0f} TF/s | gen4 {res[4][1]} {res[4][0]:.3f} ms {fl / res[4][0] / 1e9:.0f} TF/s | rel-L2 {d:.2e} | A from L2: gen3 {fl / al[3] / 1e9:.0f} gen4 {fl / al[4] / 1e9:.0f} TF/s", flush=True)
