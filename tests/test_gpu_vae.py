"""-m gpu: the HIP temporal VAE (evoworld_amd.vae, row N1) against the fp32 oracle restatement (oracle/vae_ref.py) on the
same seeded weights: encode (latent_dist.mode()) and decode (chunks with num_frames, incl. the frame-axis convs and
time_conv_out), plus its kernels (row softmax, 3-tap frame conv, asymmetric-padding stride-2 conv) against torch fp32.
Stated tolerance: rel-L2 <= 3e-3 (fp16 MFMA operands, split-fp16 residual stream, fp32 softmax); measured values are printed."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _g(s):
    return torch.Generator().manual_seed(s)


def test_softmax_rows_split_scores():
    from evoworld_amd import ops
    R, C = 300, 1024
    s = torch.randn(R, C, generator=_g(0)) * 6
    sr = ops.Res.from_float(s.to(DEV))
    hi = sr.hi
    p = ops.softmax_rows(sr)
    ref = torch.softmax(sr.float(), dim=-1)
    assert rel_l2(p.float().cpu(), ref.cpu()) < 4e-4
    p1 = ops.softmax_rows(hi)
    assert rel_l2(p1.float().cpu(), torch.softmax(hi.float(), -1).cpu()) < 4e-4


def test_time_conv3():
    from evoworld_amd import ops
    B, T, C, H, W = 2, 5, 3, 8, 12
    x = torch.randn(B, T, C, H, W, generator=_g(1))
    w, b = torch.randn(C, C, 3, generator=_g(2)), torch.randn(C, generator=_g(3))
    y = ops.time_conv3(x.to(DEV), w.to(DEV), b.to(DEV))
    ref = F.conv3d(x.permute(0, 2, 1, 3, 4), w[:, :, :, None, None], b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4)
    assert torch.allclose(y.cpu(), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("H,W", [(16, 32), (10, 14)])
def test_conv_stride2_asymmetric_padding(H, W):
    """diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then 3x3 stride 2 -- the conv_shift tap origin"""
    from evoworld_amd import ops
    N, C, O = 2, 64, 128
    x = torch.randn(N, C, H, W, generator=_g(4))
    w = torch.randn(O, C, 3, 3, generator=_g(5)) / math.sqrt(C * 9)
    b = torch.randn(O, generator=_g(6))
    xh = x.permute(0, 2, 3, 1).reshape(-1, C).half().to(DEV).contiguous()
    Ho, Wo = H // 2, W // 2
    out = torch.empty(N * Ho * Wo, O, dtype=torch.float16, device=DEV)
    ops.gemm(xh, ops.pack_conv_weight(w.to(DEV)), out, M=N * Ho * Wo, N=O, c1=C, lda=C, bias=b.half().to(DEV), mode=ops.A_CONV3X3,
             conv=(N, H, W, Ho, Wo, 2, 0), conv_shift=1)
    xr = xh.float().reshape(N, H, W, C).permute(0, 3, 1, 2)
    ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), w.half().float().to(DEV), b.half().float().to(DEV), stride=2)
    assert ref.shape[-2:] == (Ho, Wo)
    got = out.float().reshape(N, Ho, Wo, O).permute(0, 3, 1, 2)
    assert rel_l2(got.cpu(), ref.cpu()) < 1e-3


@pytest.fixture(scope="module")
def vaes():
    from evoworld_amd.vae import AutoencoderKLTemporalDecoder, DEFAULT_VAE_CONFIG, random_vae_state_dict
    from oracle.vae_ref import AutoencoderKLTemporalDecoderRef, tiny_vae_config
    cfg = tiny_vae_config()
    sd = {k: v.half().float() for k, v in random_vae_state_dict({**DEFAULT_VAE_CONFIG, **cfg}, 0).items()}
    ref = AutoencoderKLTemporalDecoderRef(**cfg).eval()
    ref.load_state_dict(sd, strict=True)                      # same key names: the diffusers layout
    vae = AutoencoderKLTemporalDecoder(**cfg).load_state_dict(sd, device=DEV)
    return cfg, ref, vae


def test_vae_encode_mode_vs_oracle(vaes):
    cfg, ref, vae = vaes
    x = torch.rand(3, 3, 64, 128, generator=_g(7)) * 2 - 1
    want = ref.encode_mode(x)
    got = vae.encode(x.to(DEV)).latent_dist.mode()
    assert got.shape == (3, 4, 8, 16)
    e = rel_l2(got.cpu(), want)
    print(f"VAE encode (mode) rel-L2 {e:.3e}")
    assert e < 3e-3
    assert torch.equal(got, vae.encode(x.to(DEV)).latent_dist.mode())      # deterministic


@pytest.mark.parametrize("n,T", [(4, 4), (6, 3)])
def test_vae_decode_vs_oracle(vaes, n, T):
    cfg, ref, vae = vaes
    z = torch.randn(n, 4, 8, 16, generator=_g(8))
    want = ref.decode(z, T)
    got = vae.decode(z.to(DEV), num_frames=T).sample
    assert got.shape == (n, 3, 64, 128)
    e = rel_l2(got.cpu(), want)
    print(f"VAE decode n={n} T={T} rel-L2 {e:.3e}")
    assert e < 3e-3


def test_vae_behind_the_pipeline_duck_type(vaes):
    """pipeline_evoworld.py:307-328,358-385: encode 1+T frames -> conditioning; decode in chunks of decode_chunk_size"""
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from evoworld_amd.unet import DEFAULT_CONFIG, UNetSpatioTemporalConditionModel, random_state_dict
    from oracle.unet_ref import tiny_config
    cfg, ref, vae = vaes
    ucfg = tiny_config()
    unet = UNetSpatioTemporalConditionModel(**ucfg).load_state_dict(random_state_dict({**DEFAULT_CONFIG, **ucfg}, 0), device=DEV)
    pipe = StableVideoDiffusionPipeline(unet=unet, vae=vae)
    T, H, W = 4, 64, 128
    lat = torch.randn(1, T, 4, H // 8, W // 8, generator=_g(9))
    frames = pipe.decode_latents(lat.to(DEV) * vae.config.scaling_factor, T, decode_chunk_size=3)    # chunks 3 + 1
    want = torch.cat([ref.decode(lat[0, :3], 3), ref.decode(lat[0, 3:], 1)]).reshape(1, T, 3, H, W).permute(0, 2, 1, 3, 4)
    assert frames.shape == (1, 3, T, H, W)
    assert rel_l2(frames.cpu(), want) < 3e-3
