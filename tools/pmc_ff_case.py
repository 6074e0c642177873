"""Three launches each of the fused level-0 feed-forward (spatial form: bias + one split residual) and of the level-2 GEGLU up-projection on
generation 3, for counter runs (tools/pmc_run.sh tools/pmc_ff_case.py ff320_kernel gemm3_kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evoworld_amd import ops
g = torch.Generator().manual_seed(0)
rows, C, HID = 460800, 320, 1280
x = (torch.rand(rows, C, generator=g) * 2 - 1).half().cuda()
w1 = (torch.randn(2 * HID, C, generator=g) * 0.05)
b1 = torch.randn(2 * HID, generator=g) * 0.1
w2 = (torch.randn(C, HID, generator=g) * 0.03)
b2 = (torch.randn(C, generator=g) * 0.1).half().cuda()
pack = ops.ff_pack(w1.cuda(), b1.cuda(), w2.cuda())
r1 = ops.Res.from_float((torch.rand(rows, C, generator=g) * 2 - 1).cuda())
out = ops.Res.empty(rows, C, "cuda", True)
M, N, K = 28800, 10240, 1280
xa = (torch.rand(M, K, generator=g) * 2 - 1).half().cuda()
wa = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).half().cuda()
ba = torch.rand(N, generator=g).half().cuda()
oa = torch.empty(M, N // 2, dtype=torch.float16, device="cuda")
for _ in range(3):
    ops.ff_geglu320(x, pack, b2, out, r1=r1)
    ops.gemm(xa, wa, oa, M=M, N=N, c1=K, lda=K, bias=ba, act=2)
torch.cuda.synchronize()
