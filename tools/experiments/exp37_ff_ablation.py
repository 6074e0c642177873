"""Round 3: where does the fused level-0 feed-forward kernel spend its time?  Times ew_ff_geglu320_f16 of the library named by
EW_LIB_PATH (ablation builds -DFF_ABL=n: 1 no GEGLU math, 2 no DMA, 4 no epilogue, 8 no up-projection MFMA, 16 no down-projection
MFMA); results of ablated builds are wrong by design."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import ops  # noqa: E402

M, C = 460800, 320
g = torch.Generator().manual_seed(0)
w1 = ((torch.rand(2560, C, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
b1 = ((torch.rand(2560, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
w2 = ((torch.rand(C, 1280, generator=g) * 2 - 1) / 1280 ** 0.5).half().cuda()
b2 = ((torch.rand(C, generator=g) * 2 - 1) / 36).half().cuda()
x = torch.randn(M, C, generator=g).half().cuda()
h = ops.Res.from_float(torch.randn(M, C, generator=g).cuda())
pack = ops.ff_pack(w1, b1, w2)
out = ops.Res.empty(M, C, "cuda", True)
fn = lambda: ops.ff_geglu320(x, pack, b2, out, r1=h)
fn(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fn()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 5)
print(f"{os.path.basename(os.environ.get('EW_LIB_PATH', 'libevoworld_hip.so')):32s} fused feed-forward 460800 tokens: {best:.3f} ms", flush=True)
