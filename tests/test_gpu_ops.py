"""-m gpu: every HIP kernel (through the C ABI) against a plain PyTorch fp32 reference of the same op.
Tolerances: fp16 storage + fp32 accumulation -> rel-L2 <= 2e-3 per op (stated per test)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _g(seed):
    return torch.Generator(device="cpu").manual_seed(seed)


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=_g(seed)) * scale)


@pytest.fixture(scope="module")
def ops():
    from evoworld_amd import ops as o
    assert torch.cuda.is_available()
    return o


# ----------------------------------------------------------------------------- GEMM (dense)
@pytest.mark.parametrize("M,N,K", [(128, 160, 64), (300, 320, 320), (257, 128, 192), (50, 1280, 1024), (2, 640, 128),
                                   (1000, 4, 320), (384, 960, 640)])
def test_gemm_dense_bias(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3)
    xh, wh, bh = x.half().to(DEV), w.half().to(DEV), b.half().to(DEV)
    out = ops.linear(xh, wh, bh)
    ref = xh.float() @ wh.float().T + bh.float()
    assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-3


def test_gemm_asymmetric_identity(ops):
    # A = I with an asymmetric W catches row/col swaps in the fragment layout
    K = 128
    x = torch.eye(K).half().to(DEV)
    w = (torch.arange(160 * K).reshape(160, K) % 251).float().div(251).half().to(DEV)
    out = ops.linear(x, w)
    assert torch.equal(out, w.T.contiguous())


def test_gemm_epilogues(ops):
    M, N, K, G = 520, 320, 256, 4
    rpg = 130
    x, w = rnd(M, K, seed=1).half().to(DEV), (rnd(N, K, seed=2) / 16).half().to(DEV)
    b, rb = rnd(N, seed=3).half().to(DEV), rnd(G, N + 64, seed=4).half().to(DEV)
    r1, r2 = rnd(M, N, seed=5).half().to(DEV), rnd(M, N, seed=6).half().to(DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, rowbias=rb[:, 64:], ld_rowbias=N + 64, rows_per_group=rpg,
             r1=r1, ld_r1=N, r2=r2, ld_r2=N, c_acc=0.4, c_r1=0.6, c_r2=-1.5)
    grp = torch.arange(M, device=DEV) // rpg
    ref = 0.4 * (x.float() @ w.float().T + b.float() + rb[:, 64:].float()[grp]) + 0.6 * r1.float() - 1.5 * r2.float()
    assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-3
    out2 = ops.linear(x, w, b, act=ops.ACT_SILU)
    assert rel_l2(out2.float().cpu(), F.silu(x.float() @ w.float().T + b.float()).cpu()) < 1e-3


def test_gemm_geglu(ops):
    M, C = 300, 64
    x = rnd(M, C, seed=1).half().to(DEV)
    w, b = (rnd(8 * C, C, seed=2) / 8).half().to(DEV), rnd(8 * C, seed=3).half().to(DEV)
    n = 4 * C
    idx = torch.arange(2 * n).reshape(2, n // 16, 16).permute(1, 0, 2).reshape(-1).to(DEV)
    out = ops.linear(x, w[idx].contiguous(), b[idx].contiguous(), act=ops.ACT_GEGLU)
    y = x.float() @ w.float().T + b.float()
    ref = y[:, :n] * F.gelu(y[:, n:])
    assert out.shape == (M, n)
    assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-3


def test_gemm_dual_source_and_transposed_out(ops):
    M, c1, c2, N = 260, 128, 64, 160
    a, a2 = rnd(M, c1, seed=1).half().to(DEV), rnd(M, c2, seed=2).half().to(DEV)
    w = (rnd(N, c1 + c2, seed=3) / 12).half().to(DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a, w, out, M=M, N=N, c1=c1, lda=c1, a2=a2, c2=c2, lda2=c2)
    ref = torch.cat([a, a2], 1).float() @ w.float().T
    assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-3
    # swapped operands: V^T = W_v X^T
    wv, x = (rnd(128, 128, seed=4) / 11).half().to(DEV), rnd(1000, 128, seed=5).half().to(DEV)
    vt = torch.empty(128, 1000, dtype=torch.float16, device=DEV)
    ops.gemm(wv, x, vt, M=128, N=1000, c1=128, lda=128)
    assert rel_l2(vt.float().cpu(), (wv.float() @ x.float().T).cpu()) < 1e-3


# ----------------------------------------------------------------------------- implicit-GEMM convs
def _nhwc(x):  # [N,C,H,W] -> [N*H*W, C] fp16
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).half().contiguous().to(DEV)


def _pack3(w):  # [O,I,3,3] -> [O, K] in the kernel's chunk-major / tap-minor K order
    from evoworld_amd.ops import pack_conv_weight
    return pack_conv_weight(w.half().float()).to(DEV)


@pytest.mark.parametrize("N,C,O,H,W,stride,up", [(3, 64, 160, 9, 16, 1, 0), (2, 128, 128, 18, 32, 2, 0),
                                                  (2, 64, 320, 5, 8, 1, 1), (1, 192, 64, 72, 128, 1, 0)])
def test_conv3x3(ops, N, C, O, H, W, stride, up):
    x, w, b = rnd(N, C, H, W, seed=1), rnd(O, C, 3, 3, seed=2) / math.sqrt(9 * C), rnd(O, seed=3)
    xh, wh, bh = x.half().float(), w.half().float(), b.half().float()
    xin = F.interpolate(xh, scale_factor=2.0, mode="nearest") if up else xh
    ref = F.conv2d(xin.to(DEV), wh.to(DEV), bh.to(DEV), stride=stride, padding=1)
    Ho, Wo = ref.shape[-2:]
    out = torch.empty(N * Ho * Wo, O, dtype=torch.float16, device=DEV)
    ops.gemm(_nhwc(x), _pack3(w), out, M=N * Ho * Wo, N=O, c1=C, lda=C, bias=b.half().to(DEV), mode=ops.A_CONV3X3,
             conv=(N, H, W, Ho, Wo, stride, up))
    got = out.float().reshape(N, Ho, Wo, O).permute(0, 3, 1, 2)
    assert rel_l2(got.cpu(), ref.cpu()) < 1e-3


def test_conv3x3_concat(ops):
    N, c1, c2, O, H, W = 2, 128, 64, 128, 10, 12
    x1, x2 = rnd(N, c1, H, W, seed=1), rnd(N, c2, H, W, seed=2)
    w = rnd(O, c1 + c2, 3, 3, seed=3) / 40
    ref = F.conv2d(torch.cat([x1, x2], 1).half().float().to(DEV), w.half().float().to(DEV), padding=1)
    out = torch.empty(N * H * W, O, dtype=torch.float16, device=DEV)
    ops.gemm(_nhwc(x1), _pack3(w), out, M=N * H * W, N=O, c1=c1, lda=c1, a2=_nhwc(x2), c2=c2, lda2=c2,
             mode=ops.A_CONV3X3, conv=(N, H, W, H, W, 1, 0))
    assert rel_l2(out.float().reshape(N, H, W, O).permute(0, 3, 1, 2).cpu(), ref.cpu()) < 1e-3


@pytest.mark.parametrize("B,T,P,C", [(2, 25, 40, 64), (1, 4, 300, 128), (2, 1, 64, 64)])
def test_conv_temporal(ops, B, T, P, C):
    x = rnd(B, T, P, C, seed=1)
    w, b = rnd(C, C, 3, 1, 1, seed=2) / math.sqrt(3 * C), rnd(C, seed=3)
    xr = x.half().float().permute(0, 3, 1, 2).unsqueeze(-1)           # [B,C,T,P,1]
    ref = F.conv3d(xr.to(DEV), w.half().float().to(DEV), b.half().float().to(DEV), padding=(1, 0, 0))
    wp = _pack3(w)
    out = torch.empty(B * T * P, C, dtype=torch.float16, device=DEV)
    ops.gemm(x.reshape(-1, C).half().to(DEV), wp, out, M=B * T * P, N=C, c1=C, lda=C, bias=b.half().to(DEV),
             mode=ops.A_CONVT3, tconv=(B, T, P))
    got = out.float().reshape(B, T, P, C).permute(0, 3, 1, 2).unsqueeze(-1)
    assert rel_l2(got.cpu(), ref.cpu()) < 1e-3


# ----------------------------------------------------------------------------- norms
@pytest.mark.parametrize("n,rows,Cs,silu", [(5, 144, [320], True), (2, 3600, [64], False), (3, 200, [1280, 640], True),
                                             (2, 77, [640, 320], True), (1, 4096, [2560], False)])
def test_groupnorm(ops, n, rows, Cs, silu):
    xs = [rnd(n * rows, c, seed=10 + i, scale=1.5) + 0.3 for i, c in enumerate(Cs)]
    C = sum(Cs)
    g, b = rnd(C, seed=3) * 0.2 + 1, rnd(C, seed=4) * 0.1
    xh = [x.half().to(DEV) for x in xs]
    out = ops.groupnorm(xh, g.half().to(DEV), b.half().to(DEV), n, rows, 1e-5, silu)
    xc = torch.cat([x.float() for x in xh], 1).reshape(n, rows, C).permute(0, 2, 1)
    ref = F.group_norm(xc, 32, g.half().float().to(DEV), b.half().float().to(DEV), 1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1).reshape(n * rows, C)
    assert rel_l2(out.float().cpu(), ref.cpu()) < 2e-3


@pytest.mark.parametrize("ratio", [30.0, 300.0])
def test_groupnorm_high_mean_over_std_and_determinism(ops, ratio):
    """channels with mean >> std: shifted sums, fixed-order reduction -> bit-identical reruns"""
    n, rows, C = 3, 1152, 640
    x = torch.randn(n * rows, C, generator=_g(20)) * 0.25 + ratio * 0.25
    g, b = rnd(C, seed=3) * 0.2 + 1, rnd(C, seed=4) * 0.1
    xh = x.half().to(DEV)
    out = ops.groupnorm([xh], g.half().to(DEV), b.half().to(DEV), n, rows, 1e-6, False)
    for _ in range(3):
        assert torch.equal(out, ops.groupnorm([xh], g.half().to(DEV), b.half().to(DEV), n, rows, 1e-6, False))
    xc = xh.double().reshape(n, rows, C).permute(0, 2, 1)
    ref = F.group_norm(xc, 32, g.half().double().to(DEV), b.half().double().to(DEV), 1e-6).permute(0, 2, 1).reshape(n * rows, C)
    assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-3


def test_groupnorm_split_stream_concat(ops):
    """two sources (virtual concat), both split fp16: statistics and apply see hi + lo"""
    n, rows, Cs = 2, 300, [640, 320]
    xs = [rnd(n * rows, c, seed=30 + i, scale=1.5) + 0.3 for i, c in enumerate(Cs)]
    C = sum(Cs)
    g, b = rnd(C, seed=3) * 0.2 + 1, rnd(C, seed=4) * 0.1
    src, full = [], []
    for x in xs:
        src.append(ops.Res.from_float(x.to(DEV)))
        full.append(src[-1].float())
    out = ops.groupnorm(src, g.half().to(DEV), b.half().to(DEV), n, rows, 1e-5, True)
    xc = torch.cat(full, 1).reshape(n, rows, C).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xc, 32, g.half().float().to(DEV), b.half().float().to(DEV), 1e-5)).permute(0, 2, 1).reshape(n * rows, C)
    e_split = rel_l2(out.float().cpu(), ref.cpu())
    out_hi = ops.groupnorm([s.hi for s in src], g.half().to(DEV), b.half().to(DEV), n, rows, 1e-5, True)
    e_hi = rel_l2(out_hi.float().cpu(), ref.cpu())
    assert e_split < 4e-4 and e_split < e_hi      # output rounding only (2^-11/sqrt(3) = 2.8e-4) vs input rounding on top


def test_groupnorm_split_operand_output_and_three_block_gemm(ops):
    """Round 6 (ew_groupnorm_apply_split_f16): the GroupNorm result as the split operand [y_hi | y_lo]: y_hi is bit-identical to the plain apply,
    y_hi + y_lo reproduces the fp32 GroupNorm to ~2^-21, and the consumer form -- weights [W_hi | W_hi | W_lo], source 2 = the y_hi half of the
    same rows -- reproduces y W^T with fp32 operands to ~1e-6 where single-rounded operands sit at ~4e-4 (the conv_out / level-0 proj_in path)."""
    n, rows, C, N = 2, 700, 320, 320
    x = rnd(n * rows, C, seed=40, scale=1.5) + 0.3
    g, b = rnd(C, seed=3) * 0.2 + 1, rnd(C, seed=4) * 0.1
    src = ops.Res.from_float(x.to(DEV))
    gh, bh = g.half().to(DEV), b.half().to(DEV)
    plain = ops.groupnorm([src], gh, bh, n, rows, 1e-6, True)
    sp = ops.groupnorm([src], gh, bh, n, rows, 1e-6, True, split_out=True)
    assert sp.shape == (n * rows, 2 * C) and torch.equal(sp[:, :C], plain)
    xc = src.float().double().reshape(n, rows, C).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xc, 32, gh.double(), bh.double(), 1e-6)).permute(0, 2, 1).reshape(n * rows, C)
    e_hi, e_split = rel_l2(plain.double().cpu(), ref.cpu()), rel_l2((sp[:, :C].double() + sp[:, C:].double()).cpu(), ref.cpu())
    print(f"GroupNorm output vs fp64: fp16 {e_hi:.2e}, split operand {e_split:.2e}")
    assert e_hi > 1e-4 and e_split < 2e-6
    w = (rnd(N, C, seed=41) / 18).to(DEV)
    w_hi = w.half().float()
    w3 = torch.cat([w_hi, w_hi, w - w_hi], 1).half().contiguous()
    out = ops.Res.empty(n * rows, N, DEV, True)
    ops.gemm(sp, w3, out, M=n * rows, N=N, c1=2 * C, lda=2 * C, a2=sp, c2=C, lda2=2 * C)
    want = ref.to(DEV) @ w.double().t()
    out1 = ops.Res.empty(n * rows, N, DEV, True)
    ops.gemm(plain, w.half().contiguous(), out1, M=n * rows, N=N, c1=C, lda=C)
    e3, e1 = rel_l2(out.float().double().cpu(), want.cpu()), rel_l2(out1.float().double().cpu(), want.cpu())
    print(f"y W^T vs fp64: single-rounded operands {e1:.2e}, split operands (3 K blocks) {e3:.2e}")
    assert e1 > 1e-4 and e3 < 3e-6


def test_gemm_split_residual_epilogues(ops):
    """r1 / r2 / out as hi + lo pairs through every kernel generation and both tile families (N = 320k and N = 128k)"""
    from evoworld_amd import _lib
    lib = _lib.load()
    for gen in (3, 2, 1):
        lib.ew_set_gemm_generation(gen)
        try:
            for (M, N, K) in ((1100, 640, 256), (1100, 320, 192), (300, 384, 128)):
                x, w = rnd(M, K, seed=1).half().to(DEV), (rnd(N, K, seed=2) / 16).half().to(DEV)
                b = rnd(N, seed=3).half().to(DEV)
                r1f, r2f = rnd(M, N, seed=5) * 3, rnd(M, N, seed=6) * 3
                r1h, r2h = r1f.half().to(DEV), r2f.half().to(DEV)
                r1, r2 = ops.Res.from_float(r1f.to(DEV)), ops.Res.from_float(r2f.to(DEV))
                assert torch.equal(r1.hi, r1h) and torch.equal(r2.hi, r2h)          # hi is the plain fp16 rounding
                assert rel_l2(r1.float().cpu(), r1f) < 2e-6                         # ~19 mantissa bits in 3 bytes
                out = ops.Res.empty(M, N, DEV, True)
                ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, r1=r1, ld_r1=N, r2=r2, ld_r2=N, c_acc=0.4, c_r1=0.6, c_r2=-1.5)
                ref = 0.4 * (x.float() @ w.float().T + b.float()) + 0.6 * r1.float() - 1.5 * r2.float()
                assert rel_l2(out.float().cpu(), ref.cpu()) < 2e-5, (gen, M, N, K)
                # r1 only, lo on the input only (the blended hb of the transformer: fp16 out)
                o2 = torch.empty(M, N, dtype=torch.float16, device=DEV)
                ops.gemm(x, w, o2, M=M, N=N, c1=K, lda=K, bias=b, r1=r1, ld_r1=N)
                ref2 = x.float() @ w.float().T + b.float() + r1.float()
                assert rel_l2(o2.float().cpu(), ref2.cpu()) < 4e-4, (gen, M, N, K)
                # no residual, lo on the output only (proj_in / conv_in)
                o3 = ops.Res.empty(M, N, DEV, True)
                ops.gemm(x, w, o3, M=M, N=N, c1=K, lda=K, bias=b)
                assert rel_l2(o3.float().cpu(), (x.float() @ w.float().T + b.float()).cpu()) < 2e-5, (gen, M, N, K)
        finally:
            lib.ew_set_gemm_generation(3)


@pytest.mark.parametrize("rows,C", [(37, 320), (1000, 640), (9, 1280), (130, 64)])
def test_layernorm(ops, rows, C):
    x, g, b = rnd(rows, C, seed=1) * 2 + 0.5, rnd(C, seed=2) * 0.2 + 1, rnd(C, seed=3) * 0.1
    xh, gh, bh = x.half().to(DEV), g.half().to(DEV), b.half().to(DEV)
    out = ops.layernorm(xh, gh, bh)
    ref = F.layer_norm(xh.float(), (C,), gh.float(), bh.float(), 1e-5)
    assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-3
    rpg = 7
    av = rnd((rows + rpg - 1) // rpg, C, seed=4).half().to(DEV)
    xo = torch.empty_like(xh)
    out2 = ops.layernorm(xh, gh, bh, addvec=av, rows_per_group=rpg, x_out=xo)
    xs = (xh.float() + av.float()[torch.arange(rows, device=DEV) // rpg]).half()
    assert torch.equal(xo, xs)
    assert rel_l2(out2.float().cpu(), F.layer_norm(xs.float(), (C,), gh.float(), bh.float(), 1e-5).cpu()) < 1e-3
    # split stream (hi, lo8) in and out
    xs3 = ops.Res.from_float(x.to(DEV))
    xo3 = ops.Res.empty(rows, C, DEV, True)
    out3 = ops.layernorm(xs3, gh, bh, addvec=av, rows_per_group=rpg, x_out=xo3)
    full = xs3.float() + av.float()[torch.arange(rows, device=DEV) // rpg]
    assert rel_l2(xo3.float().cpu(), full.cpu()) < 2e-6
    assert rel_l2(out3.float().cpu(), F.layer_norm(full, (C,), gh.float(), bh.float(), 1e-5).cpu()) < 4e-4


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize("n_seq,S,heads", [(2, 8, 1), (3, 144, 2), (2, 576, 3), (1, 2304, 2), (2, 200, 1)])
def test_attn_spatial(ops, n_seq, S, heads):
    C = heads * 64
    rows = n_seq * S
    qk = rnd(rows, 2 * C, seed=1).half().to(DEV)
    v = rnd(rows, C, seed=2).half().to(DEV)
    qk[: S // 2, :64] *= 4.0   # sharpen some rows so the running max moves between tiles
    vt = v.T.contiguous()
    o = torch.empty(rows, C, dtype=torch.float16, device=DEV)
    ops.attn_spatial(qk, qk[:, C:], vt, o, n_seq, S, heads, 2 * C, rows, C)
    q = qk[:, :C].float().reshape(n_seq, S, heads, 64).transpose(1, 2)
    k = qk[:, C:].float().reshape(n_seq, S, heads, 64).transpose(1, 2)
    vv = v.float().reshape(n_seq, S, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(q, k, vv).transpose(1, 2).reshape(rows, C)
    assert torch.isfinite(o).all()
    assert rel_l2(o.float().cpu(), ref.cpu()) < 2e-3


@pytest.mark.parametrize("n_seq,S,heads,shift", [(2, 512, 2, 0.0), (1, 1000, 1, 0.0), (3, 136, 5, 0.0), (1, 2304, 1, -60.0), (1, 640, 2, 40.0)])
def test_attn_spatial_log2(ops, n_seq, S, heads, shift):
    """ew_attn_spatial_log2_f16: q, k pre-scaled by sqrt(scale * log2 e) (what the projection epilogue writes); the MFMA's C operand
    subtracts the running max.  `shift` moves every score of a head by a constant (one extra q / k channel pair): strongly negative
    first-tile maxima (the max must be SET on the first tile, not only raised) and large positive ones (deferred-max raises)."""
    C = heads * 64
    rows = n_seq * S
    qk32 = rnd(rows, 2 * C, seed=1)
    v = rnd(rows, C, seed=2).half().to(DEV)
    qk32[: S // 2, :64] *= 4.0                     # sharpen some rows so the running max moves between tiles
    if shift:
        qk32[:, 0] = abs(shift) ** 0.5 * 8 ** 0.5  # q_0 * k_0 * scale = shift
        qk32[:, C] = (1 if shift > 0 else -1) * abs(shift) ** 0.5 * 8 ** 0.5
        qk32[S // 3:, C] *= 0.5                    # ... and smaller for later keys: the first tile holds the max
    qk = (qk32 * ops.QK_LOG2_PRESCALE).half().to(DEV)
    vt = v.T.contiguous()
    o = torch.empty(rows, C, dtype=torch.float16, device=DEV)
    ops.attn_spatial_log2(qk, qk[:, C:], vt, o, n_seq, S, heads, 2 * C, rows, C)
    q = qk[:, :C].float().reshape(n_seq, S, heads, 64).transpose(1, 2)
    k = qk[:, C:].float().reshape(n_seq, S, heads, 64).transpose(1, 2)
    vv = v.float().reshape(n_seq, S, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(q, k, vv, scale=math.log(2.0)).transpose(1, 2).reshape(rows, C)   # 2^(q.k) = e^(ln2 q.k)
    assert torch.isfinite(o).all()
    e = rel_l2(o.float().cpu(), ref.cpu())
    print(f"attn_spatial_log2 n_seq={n_seq} S={S} heads={heads} shift={shift}: rel-L2 {e:.2e}")
    assert e < 2e-3


@pytest.mark.parametrize("pattern", ["ramp", "jumps", "overflow", "flat_below", "flat_above", "late_spike", "ragged_spike"])
def test_attn_spatial_log2_lazy_max(ops, pattern):
    """Round 5: the log2 kernel looks at a tile's maximum only when the tile's ROW SUM leaves the safe range (and on the first tile).  Score
    profiles that stress exactly that decision: key j carries an offset f(j) in log2 units (q channel 0 = 1, k channel 0 = f) on top of random
    scores -- slow ramps (many small raises), jumps up and down, jumps large enough to overflow the exponentials before the check sees them,
    plateaus just below / just above the row-sum limit (32 keys x 2^4.9 = 955 < 1024 < 32 x 2^5.1), a spike in the very last (ragged) tile."""
    n_seq, heads = 2, 2
    S = 1000 if pattern == "ragged_spike" else 1024
    C = heads * 64
    rows = n_seq * S
    qk32 = rnd(rows, 2 * C, seed=5) * 0.6
    v = rnd(rows, C, seed=6).half().to(DEV)
    t = (torch.arange(S) // 64).float()
    f = {"ramp": 0.37 * t, "jumps": 30.0 * ((t % 3) == 2).float() - 12.0 * ((t % 5) == 1).float(), "overflow": 200.0 * t,
         "flat_below": 4.9 * (t > 0).float(), "flat_above": 5.1 * (t > 0).float(), "late_spike": 25.0 * (t == 15).float(),
         "ragged_spike": 40.0 * (torch.arange(S) >= 990).float()}[pattern]
    for h in range(heads):
        qk32[:, h * 64] = 1.0
        qk32[:, C + h * 64] = f.repeat(n_seq) * (1.0 if h == 0 else -0.5)       # the second head sees the mirrored (falling) profile
    qk = qk32.half().to(DEV)                          # already in log2 units: what ew_attn_spatial_log2_f16 takes
    vt = v.T.contiguous()
    o = torch.empty(rows, C, dtype=torch.float16, device=DEV)
    ops.attn_spatial_log2(qk, qk[:, C:], vt, o, n_seq, S, heads, 2 * C, rows, C)
    q = qk[:, :C].double().reshape(n_seq, S, heads, 64).transpose(1, 2)
    k = qk[:, C:].double().reshape(n_seq, S, heads, 64).transpose(1, 2)
    vv = v.double().reshape(n_seq, S, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(q, k, vv, scale=math.log(2.0)).transpose(1, 2).reshape(rows, C)
    assert torch.isfinite(o).all()
    e = rel_l2(o.double().cpu(), ref.cpu())
    print(f"attn_spatial_log2 lazy max, {pattern}: rel-L2 {e:.2e}")
    assert e < 2e-3


@pytest.mark.parametrize("B,T,S,heads", [(2, 25, 37, 2), (1, 4, 512, 1), (2, 1, 9, 3), (1, 32, 5, 1), (2, 49, 21, 2), (1, 64, 7, 1),
                                         (1, 33, 130, 3)])
def test_attn_temporal(ops, B, T, S, heads):
    C = heads * 64
    rows = B * T * S
    qkv = rnd(rows, 3 * C, seed=1).half().to(DEV)
    o = torch.empty(rows, C, dtype=torch.float16, device=DEV)
    ops.attn_temporal(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B, T, S, heads, 3 * C, C)

    def split(i):  # [B,T,S,h,64] -> [B,S,h,T,64]
        return qkv[:, i * C:(i + 1) * C].float().reshape(B, T, S, heads, 64).permute(0, 2, 3, 1, 4)
    ref = F.scaled_dot_product_attention(split(0), split(1), split(2))            # [B,S,h,T,64]
    ref = ref.permute(0, 3, 1, 2, 4).reshape(rows, C)
    assert rel_l2(o.float().cpu(), ref.cpu()) < 2e-3


# ----------------------------------------------------------------------------- glue
def test_sinusoid_embed_matches_torch_expression(ops):
    """ew_sinusoid_embed_f16 against the torch expression of diffusers' Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) it replaces in
    the forward (timestep broadcast over the batch rows; the three added time ids, one row each)."""
    import math
    for vals, rows, dim in (([1.6377], 2, 320), ([6.0, 127.0, 0.02, 6.0, 127.0, 0.02], 6, 256), ([0.0, 1.0, 24.0], 3, 64), ([-0.7], 1, 1280)):
        v = torch.tensor(vals, dtype=torch.float32)
        got = ops.sinusoid_embed(v.to(DEV), rows, dim)
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        args = v[torch.arange(rows) % len(vals)].reshape(-1, 1) * freqs[None]
        want = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        assert got.shape == (rows, dim) and got.dtype == torch.float16
        assert float((got.float().cpu() - want).abs().max()) < 6e-4       # half an fp16 ulp at 1.0 plus fp32 trig differences at |arg| <= 127


def test_layout_roundtrip(ops):
    x = rnd(3, 18, 8, 16, seed=1)
    y = torch.zeros(3 * 8 * 16, 64, dtype=torch.float16, device=DEV)
    ops.nchw_f32_to_nhwc_f16(x.to(DEV), y, 64, c_off=0, scale=0.5)
    ref = (0.5 * x).permute(0, 2, 3, 1).reshape(-1, 18).half()
    assert torch.equal(y[:, :18].cpu(), ref) and (y[:, 18:] == 0).all()
    back = ops.nhwc_f16_to_nchw_f32(y, 3, 18, 8, 16, 64)
    assert torch.equal(back.cpu(), ref.float().reshape(3, 8, 16, 18).permute(0, 3, 1, 2))


def test_layout_split_operand(ops):
    """ew_nchw_f32_to_nhwc_split_f16 (ABI 9): [x_hi | x_lo | x_hi 2^-10] channel blocks of the split model-input row, bit-exact."""
    x = rnd(3, 14, 8, 16, seed=4) * 3.0
    y = torch.full((3 * 8 * 16, 64), 5.0, dtype=torch.float16, device=DEV)
    ops.nchw_f32_to_nhwc_f16(x.to(DEV), y, 64, c_off=4, scale=0.5, split=(20, 40))
    v = (0.5 * x).permute(0, 2, 3, 1).reshape(-1, 14)
    hi = v.half()
    lo = (v - hi.float()).half()
    dup = (hi.float() * 2.0 ** -10).half()
    y = y.cpu()
    assert torch.equal(y[:, 4:18], hi) and torch.equal(y[:, 24:38], lo) and torch.equal(y[:, 44:58], dup)
    keep = torch.ones(64, dtype=torch.bool)
    keep[4:18] = keep[24:38] = keep[44:58] = False
    assert (y[:, keep] == 5.0).all()
    # hi + lo carries ~21 bits of the fp32 value
    assert rel_l2(hi.float() + lo.float(), v) < 2e-6
    with pytest.raises(Exception):
        ops.nchw_f32_to_nhwc_f16(x.to(DEV), torch.zeros(3 * 8 * 16, 64, dtype=torch.float16, device=DEV), 64, c_off=4, split=(16, 40))   # blocks overlap


def test_euler_cfg_step_split(ops):
    """ew_euler_cfg_step_split: same step; the next model input as hi / lo / hi 2^-10 blocks on both CFG rows, other channels untouched."""
    T, h, w = 5, 8, 16
    eps = rnd(2 * T * h * w, 4, seed=1).half().to(DEV)
    lat = (rnd(T, 4, h, w, seed=2) * 300).to(DEV)
    lat_b = lat.clone()
    guid = torch.linspace(1, 3, T).to(DEV)
    nxt = torch.full((2 * T * h * w, 64), 7.0, dtype=torch.float16, device=DEV)
    nxt_b = nxt.clone()
    sigma, sigma_next = 421.56912, 322.45367
    ops.euler_cfg_step(eps, 4, lat, guid, sigma, sigma_next, nxt, 64, T, h, w, split=(20, 40))
    ops.euler_cfg_step(eps, 4, lat_b, guid, sigma, sigma_next, nxt_b, 64, T, h, w)
    assert torch.equal(lat, lat_b) and torch.equal(nxt[:, :4], nxt_b[:, :4])
    v = (lat.double() / (sigma_next ** 2 + 1) ** 0.5).permute(0, 2, 3, 1).reshape(T * h * w, 4)
    got = nxt.reshape(2, T * h * w, 64)
    assert torch.equal(got[0], got[1])
    hi, lo, dup = got[0, :, :4], got[0, :, 20:24], got[0, :, 40:44]
    assert torch.equal(dup, (hi.float() * 2.0 ** -10).half())
    assert rel_l2((hi.double() + lo.double()).cpu(), v.cpu()) < 3e-6          # fp32 arithmetic of the kernel, ~21 bits kept
    keep = torch.ones(64, dtype=torch.bool)
    keep[0:4] = keep[20:24] = keep[40:44] = False
    assert (got[:, :, keep] == 7.0).all()


def test_mfma_keeps_fp16_denormal_operands(ops):
    """W = W_hi + W_lo as a second K block (conv_out, level-0 proj_in / proj_out): W_lo ~ 2^-12 |W| is a SUBNORMAL fp16 number for typical
    weights, so the split only works if the MFMA does not flush fp16 denormal inputs.  Checked through the GEMM itself."""
    M, N, K = 512, 320, 64
    a = rnd(M, K, seed=1).half().to(DEV)
    w = (rnd(N, K, seed=2) * 2e-6).half().to(DEV)                 # every element subnormal (|w| < 6.1e-5), ~30 quanta of 5.96e-8
    assert (w.float().abs() < 6.0e-5).all() and (w != 0).float().mean() > 0.9
    got = ops.linear(a, w, c_acc=4096.0).float().cpu()          # (the epilogue scale lifts the products out of fp16's own subnormal range)
    ref = (a.float().cpu() @ w.float().cpu().T) * 4096.0
    e = rel_l2(got, ref)
    print(f"GEMM with subnormal fp16 weights: rel-L2 {e:.2e} (flushed inputs would give 1.0)")
    assert e < 1e-3


def test_euler_cfg_step(ops):
    T, h, w = 5, 8, 16
    eps = rnd(2 * T * h * w, 4, seed=1).half().to(DEV)
    lat = (rnd(T, 4, h, w, seed=2) * 300).to(DEV)
    lat0 = lat.clone()
    guid = torch.linspace(1, 3, T).to(DEV)
    nxt = torch.full((2 * T * h * w, 64), 7.0, dtype=torch.float16, device=DEV)
    sigma, sigma_next = 421.56912, 322.45367
    ops.euler_cfg_step(eps, 4, lat, guid, sigma, sigma_next, nxt, 64, T, h, w)
    e = eps.float().reshape(2, T, h, w, 4).permute(0, 1, 4, 2, 3)
    e = e[0] + guid.view(T, 1, 1, 1) * (e[1] - e[0])
    s = torch.tensor(sigma)
    x0 = e * (-s / (s ** 2 + 1) ** 0.5) + lat0 / (s ** 2 + 1)
    ref = lat0 + (lat0 - x0) / s * (sigma_next - sigma)
    assert torch.allclose(lat, ref, rtol=1e-5, atol=1e-3)
    nin = (ref / (sigma_next ** 2 + 1) ** 0.5).permute(0, 2, 3, 1).reshape(T * h * w, 4)
    got = nxt.float().reshape(2, T * h * w, 64)
    assert torch.allclose(got[0, :, :4], nin, rtol=2e-3, atol=1e-3) and torch.equal(got[0], got[1])
    assert (got[:, :, 4:] == 7.0).all()


# ----------------------------------------------------------------------------- geometry (goldens from the reference)
def test_plucker_golden(ops, golden_dir):
    g = np.load(f"{golden_dir}/plucker.npz")
    for tag in ("ps01", "ps10"):
        out = ops.plucker_embed(torch.tensor(g["rays_72x128"]).to(DEV), torch.tensor(g[f"c2w_{tag}"]).to(DEV))
        assert out.shape == (25, 6, 72, 128)
        np.testing.assert_allclose(out[[0, 12, 24]].cpu().numpy(), g[f"plucker_{tag}_f0_12_24"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(out.double().sum(dim=(2, 3)).cpu().numpy(), g[f"plucker_{tag}_rowsum"], rtol=1e-6, atol=2e-3)
    out = ops.plucker_embed(torch.tensor(g["rays_8x16"]).to(DEV), torch.tensor(g["rand_c2w_rel"]).to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), g["rand_plucker_8x16"], rtol=0, atol=5e-6)


def test_cube2equi_gather_golden_bit_exact(ops, golden_dir):
    g = np.load(f"{golden_dir}/cube2equi_gather.npz")
    lut = np.load(f"{golden_dir}/cube2equi_lut.npz")["lut_64x32x16"]
    faces = torch.tensor(g["faces"]).permute(0, 1, 3, 4, 2).contiguous().to(DEV)      # [B,6,res,res,3]
    pano = ops.cube2equi_gather(faces, torch.tensor(lut).to(DEV), 32, 64)
    assert np.array_equal(pano.cpu().numpy(), g["pano"])
