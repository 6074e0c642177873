"""A/B of the short-K generation on the level-0 shapes (run once with EW_GEMM_SK=0 and once with 1)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evoworld_amd import ops, _lib
M, K = 460800, 320
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).half().cuda()
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for N, act, name in ((2560, ops.ACT_GEGLU, "GEGLU up"), (640, 0, "q|k"), (960, 0, "t q|k|v")):
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    out = torch.empty(M, N // 2 if act == ops.ACT_GEGLU else N, dtype=torch.float16, device="cuda")
    ms = t(lambda: ops.linear(x, w, b, out=out, act=act))
    print(f"SK={os.environ.get('EW_GEMM_SK','1')} {name:10s} {M}x{N}x{K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s  [{_lib.load().ew_gemm_last_kernel().decode()}]")
