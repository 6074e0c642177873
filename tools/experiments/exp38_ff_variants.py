"""Round 3: the three level-0 feed-forward call shapes of the forward (spatial ff: Res stream in / out; ff_in: + per-frame add vector,
fp16 out; temporal ff: AlphaBlender epilogue) timed as LayerNorm + two GEMMs and as the fused kernel (with LayerNorm prologue, and
with a separate LayerNorm kernel in front)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import ops  # noqa: E402
from evoworld_amd.ops import Res, ACT_GEGLU  # noqa: E402

M, C, S = 460800, 320, 18432
g = torch.Generator().manual_seed(0)
w1 = ((torch.rand(2560, C, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
b1 = ((torch.rand(2560, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
w2 = ((torch.rand(C, 1280, generator=g) * 2 - 1) / 1280 ** 0.5).half().cuda()
b2 = ((torch.rand(C, generator=g) * 2 - 1) / 36).half().cuda()
gam = (1 + 0.1 * torch.randn(C, generator=g)).half().cuda()
bet = (0.1 * torch.randn(C, generator=g)).half().cuda()
h = Res.from_float(torch.randn(M, C, generator=g).cuda())
h2 = Res.from_float(torch.randn(M, C, generator=g).cuda())
hp = torch.randn(M, C, generator=g).half().cuda()
pos = torch.randn(M // S, C, generator=g).half().cuda()
pack = ops.ff_pack(w1, b1, w2)
outR = Res.empty(M, C, "cuda", True)
outH = torch.empty(M, C, dtype=torch.float16, device="cuda")


def timeit(fn, name):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    print(f"{name:58s} {best:.3f} ms", flush=True)


def trio_spatial():
    n3 = ops.layernorm(h, gam, bet)
    ffh = ops.linear(n3, w1, b1, act=ACT_GEGLU)
    ops.linear(ffh, w2, b2, out=outR, r1=h, ld_r1=C)


def trio_ffin():
    n = ops.layernorm(h, gam, bet, addvec=pos, rows_per_group=S)
    ffh = ops.linear(n, w1, b1, act=ACT_GEGLU)
    ops.linear(ffh, w2, b2, out=outH, r1=h, ld_r1=C, rowbias=pos, rows_per_group=S, ld_rowbias=C)


def trio_temporal():
    n3 = ops.layernorm(hp, gam, bet)
    ffh = ops.linear(n3, w1, b1, act=ACT_GEGLU)
    ops.linear(ffh, w2, b2, out=outH, c_acc=0.5, r1=hp, ld_r1=C, c_r1=0.5, r2=h, ld_r2=C, c_r2=0.5)


timeit(lambda: ops.layernorm(h, gam, bet), "LayerNorm (Res in)")
timeit(lambda: ops.layernorm(hp, gam, bet), "LayerNorm (fp16 in)")
n3 = ops.layernorm(h, gam, bet)
ffh = ops.linear(n3, w1, b1, act=ACT_GEGLU)
timeit(lambda: ops.linear(n3, w1, b1, act=ACT_GEGLU), "GEGLU up-projection GEMM")
timeit(lambda: ops.linear(ffh, w2, b2, out=outR, r1=h, ld_r1=C), "down-projection GEMM + Res residual")
timeit(trio_spatial, "spatial ff : LayerNorm + 2 GEMMs")
timeit(lambda: ops.ff_geglu320(h, pack, b2, outR, r1=h, ln=(gam, bet)), "spatial ff : fused, LayerNorm prologue")
timeit(lambda: ops.ff_geglu320(ops.layernorm(h, gam, bet), pack, b2, outR, r1=h), "spatial ff : LayerNorm kernel + fused")
timeit(trio_ffin, "ff_in      : LayerNorm + 2 GEMMs")
timeit(lambda: ops.ff_geglu320(h, pack, b2, outH, r1=h, rowbias=pos, rows_per_group=S, ld_rowbias=C, ln=(gam, bet), addvec=pos,
                               add_rows_per_group=S), "ff_in      : fused, LayerNorm prologue")
timeit(lambda: ops.ff_geglu320(ops.layernorm(h, gam, bet, addvec=pos, rows_per_group=S), pack, b2, outH, r1=h, rowbias=pos, rows_per_group=S,
                               ld_rowbias=C), "ff_in      : LayerNorm kernel + fused")
timeit(trio_temporal, "temporal ff: LayerNorm + 2 GEMMs")
timeit(lambda: ops.ff_geglu320(hp, pack, b2, outH, c_acc=0.5, r1=hp, c_r1=0.5, r2=h, c_r2=0.5, ln=(gam, bet)), "temporal ff: fused, LayerNorm prologue")
timeit(lambda: ops.ff_geglu320(ops.layernorm(hp, gam, bet), pack, b2, outH, c_acc=0.5, r1=hp, c_r1=0.5, r2=h, c_r2=0.5),
       "temporal ff: LayerNorm kernel + fused")
