"""-m gpu: the optional fp8 (OCP e4m3) q / k / v projection path of BASELINE.json configs[4].
Stated tolerances: the fp8 GEMM against an fp32 matmul of the SAME quantised operands <= 1e-3 (fp16 output rounding only:
the kernel is exact on what it is given); against the unquantised fp32 product <= 4e-2 (e4m3 has 3 mantissa bits: 2^-4
relative rounding per operand); the tiny U-Net forward with fp8 q/k/v against the fp32 oracle <= 1e-2 (measured 6.4e-3)."""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _g(s):
    return torch.Generator().manual_seed(s)


def _deq(q, sc):
    return q.view(torch.float8_e4m3fn).float() * sc[:, None]


@pytest.mark.parametrize("M,N,K", [(300, 640, 320), (1000, 132, 640), (132, 1280, 1280), (64, 64, 64)])
def test_quant_and_gemm_fp8(M, N, K):
    from evoworld_amd import ops
    x = (torch.randn(M, K, generator=_g(1)) * 1.5).half().to(DEV)
    w = (torch.randn(N, K, generator=_g(2)) / math.sqrt(K)).to(DEV)
    xq, xs = ops.quant_rows_fp8(x)
    # per-row dynamic scale: amax/448, values are the e4m3 rounding of x/scale
    assert torch.allclose(xs, x.float().abs().amax(1) / 448.0, rtol=1e-6)
    assert rel_l2(_deq(xq, xs).cpu(), x.float().cpu()) < 4e-2
    ws = (w.abs().amax(1) / 448.0).contiguous()
    wq = (w / ws[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    out = ops.gemm_fp8(xq, xs, wq, ws)
    exact = _deq(xq, xs) @ _deq(wq, ws).T
    assert rel_l2(out.float().cpu(), exact.cpu()) < 1e-3                      # the kernel itself: fp16 output rounding only
    assert rel_l2(out.float().cpu(), (x.float() @ w.T).cpu()) < 4e-2          # vs the unquantised product
    # swapped roles = transposed product (how V^T is produced)
    outT = ops.gemm_fp8(wq, ws, xq, xs)
    assert rel_l2(outT.float().cpu(), exact.T.cpu()) < 1e-3


def test_gemm_fp8_asymmetric_identity():
    """A = I (exact in e4m3) against an asymmetric W: catches row/column or k-slice swaps in the fragment layout"""
    from evoworld_amd import ops
    K = 128
    eye = torch.eye(K).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV).contiguous()
    w = ((torch.arange(192 * K).reshape(192, K) % 13) - 6).float()           # small integers: exact in e4m3
    wq = w.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV).contiguous()
    one_m, one_n = torch.ones(K, device=DEV), torch.ones(192, device=DEV)
    out = ops.gemm_fp8(eye, one_m, wq, one_n)
    assert torch.equal(out.float().cpu(), w.T.contiguous())


def test_unet_tiny_fp8_qkv_vs_oracle():
    from evoworld_amd.unet import DEFAULT_CONFIG, UNetSpatioTemporalConditionModel, random_state_dict
    from oracle.unet_ref import UNetSpatioTemporalConditionModelRef, tiny_config
    cfg = tiny_config()
    sd = {k: v.half().float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}
    ref = UNetSpatioTemporalConditionModelRef(**cfg).eval()
    ref.load_state_dict(sd, strict=True)
    m = UNetSpatioTemporalConditionModel(qkv_fp8=True, **cfg).load_state_dict(sd, device=DEV)
    B, T, h, w = 2, 4, 16, 32
    g = _g(1)
    x = torch.randn(B, T, 18, h, w, generator=g)
    ehs = torch.randn(B, 1, cfg["cross_attention_dim"], generator=g)
    ehs[0] = 0
    ids = torch.tensor([[6.0, 127.0, 0.02]] * B)
    t = torch.tensor(1.6377)
    got = m(x.to(DEV), t, ehs.to(DEV), ids.to(DEV), return_dict=False)[0]
    e = rel_l2(got.cpu(), ref(x, t, ehs, ids))
    print(f"unet tiny forward with fp8 q/k/v rel-L2 {e:.3e}")
    assert torch.isfinite(got).all() and e < 1e-2


def test_fp8_weight_packs_travel_with_packed_tensors():
    """Round-2 advisor: the fp8 q/k/v packs are (weights, scales) tuples nested in the per-block dicts; `packed_tensors()` (what
    `broadcast_weights` ships and `weights_checksum` sums) must include them, or every rank but 0 keeps all-zero fp8 weights.
    Simulates the broadcast in one process: copy rank 0's packed tensors into a from_zeros replica, forwards must be identical."""
    from evoworld_amd.unet import UNetSpatioTemporalConditionModel
    from oracle.unet_ref import tiny_config
    cfg = tiny_config()
    src = UNetSpatioTemporalConditionModel.from_random(seed=3, device=DEV, qkv_fp8=True, **cfg)
    dst = UNetSpatioTemporalConditionModel.from_zeros(device=DEV, qkv_fp8=True, **cfg)
    a, b = src.packed_tensors(), dst.packed_tensors()
    assert len(a) == len(b) and any(t.dtype == torch.uint8 for t in a)          # the e4m3 bytes are in the list
    for s_, d_ in zip(a, b):
        assert s_.shape == d_.shape and s_.dtype == d_.dtype
        d_.copy_(s_)
    for k in src.w:
        if isinstance(src.w[k], dict) and "mix" in src.w[k]:
            dst.w[k]["mix"] = src.w[k]["mix"]
    assert abs(src.weights_checksum() - dst.weights_checksum()) == 0.0
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, cfg["num_frames"], 18, 16, 32, generator=g).to(DEV)
    ehs = torch.randn(2, 1, cfg["cross_attention_dim"], generator=g).to(DEV)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2).to(DEV)
    ya = src(x, torch.tensor(1.0), ehs, ids, return_dict=False)[0]
    yb = dst(x, torch.tensor(1.0), ehs, ids, return_dict=False)[0]
    assert torch.equal(ya, yb) and float(ya.abs().mean()) > 0
