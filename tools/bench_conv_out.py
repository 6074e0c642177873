"""conv_out at its forward shape (460800 pixels, 320 -> 4 channels, split operands: rows [x_hi | x_lo] + x_hi again = 8640 deep): time + check vs generation 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evoworld_amd import _lib, ops
lib = _lib.load()
N, H, W, C, O = 50, 72, 128, 320, 4
M = N * H * W
g = torch.Generator().manual_seed(0)
a = torch.randn(M, 2 * C, generator=g).half().cuda()
a[:, C:] *= 2 ** -11
w = ops.pack_conv_weight((torch.randn(O, 3 * C, 3, 3, generator=g) / 50).cuda())
b = torch.randn(O, generator=g).half().cuda()
out = torch.empty(M, O, dtype=torch.float16, device="cuda")
run = lambda: ops.gemm(a, w, out, M=M, N=O, c1=2 * C, lda=2 * C, a2=a, c2=C, lda2=2 * C, bias=b, mode=ops.A_CONV3X3, conv=(N, H, W, H, W, 1, 0))
for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    run()
e.record(); torch.cuda.synchronize()
got = out.clone()
print(lib.ew_gemm_last_kernel().decode(), f"{s.elapsed_time(e) / 10 * 1e3:.1f} us")
lib.ew_set_gemm_generation(1)
run(); torch.cuda.synchronize()
lib.ew_set_gemm_generation(3)
print("rel-L2 vs generation 1:", float((got.float() - out.float()).norm() / out.float().norm()))
