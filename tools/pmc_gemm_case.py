"""Three launches each of the level-1 GEGLU up-projection and the level-1 3x3 conv (+ split residual) for counter runs
(tools/pmc_run.sh tools/pmc_gemm_case.py gemm3_kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evoworld_amd import ops
M, N, K = 115200, 5120, 640
x = torch.rand(M, K, device="cuda", dtype=torch.float16) * 2 - 1
w = (torch.rand(N, K, device="cuda", dtype=torch.float16) * 2 - 1) * 0.05
b = torch.rand(N, device="cuda", dtype=torch.float16)
out = torch.empty(M, N // 2, dtype=torch.float16, device="cuda")
n, C, H, W = 50, 640, 36, 64
Mc = n * H * W
xc = torch.rand(Mc, C, device="cuda", dtype=torch.float16) * 2 - 1
wc = (torch.rand(C, 9 * C, device="cuda", dtype=torch.float16) * 2 - 1) * 0.02
r1 = ops.Res.from_float(torch.rand(Mc, C, device="cuda") * 2 - 1)
oc = ops.Res.empty(Mc, C, "cuda", True)
for _ in range(3):
    ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, act=2)
    ops.gemm(xc, wc, oc, M=Mc, N=C, c1=C, lda=C, bias=b[:C], mode=ops.A_CONV3X3, conv=(n, H, W, H, W, 1, 0), r1=r1, ld_r1=C)
torch.cuda.synchronize()
