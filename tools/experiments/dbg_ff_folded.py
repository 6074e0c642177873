import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from evoworld_amd import ops
DEV = "cuda"
g = lambda s: torch.Generator().manual_seed(s)
C, H = 320, 1280
w1 = ((torch.rand(2 * H, C, generator=g(1)) * 2 - 1) / C ** 0.5).half().to(DEV)
b1 = ((torch.rand(2 * H, generator=g(2)) * 2 - 1) / C ** 0.5).half().to(DEV)
w2 = ((torch.rand(C, H, generator=g(3)) * 2 - 1) / H ** 0.5).half().to(DEV)
b2 = ((torch.rand(C, generator=g(4)) * 2 - 1) / H ** 0.5).half().to(DEV)
gm, bt = (torch.rand(C, generator=g(7)) + 0.5).half().to(DEV), (torch.rand(C, generator=g(8)) - 0.5).half().to(DEV)
pack, pack_ln = ops.ff_pack(w1, b1, w2), ops.ff_pack(w1, b1, w2, ln=(gm, bt))
for M in (128, 256, 1000, 128 * 256, 128 * 300, 460800):
    hf = torch.randn(M, C, generator=g(1)) * 2 + 0.3
    h = ops.Res.from_float(hf.to(DEV))
    want, got = ops.Res.empty(M, C, DEV, True), ops.Res.empty(M, C, DEV, True)
    ops.ff_geglu320(ops.layernorm(h, gm, bt), pack, b2, want, r1=h)
    ops.ff_geglu320(h, pack_ln, b2, got, r1=h, ln_folded=True)
    torch.cuda.synchronize()
    gf, wf = got.float(), want.float()
    bad = ~torch.isfinite(gf)
    rows = bad.any(1).nonzero().flatten()
    d = (gf - wf).abs()
    d[bad] = 0
    print(f"M={M}: non-finite {int(bad.sum())} in {rows.numel()} rows (first {rows[:8].tolist()}, tiles {sorted(set((rows // 128).tolist()))[:10]}); max |diff| elsewhere {float(d.max()):.3e}; rel {float((d.norm() / wf.norm())):.2e}")
