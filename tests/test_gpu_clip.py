"""-m gpu: row N2 -- the reference's antialiased 224x224 resize and the CLIP ViT image encoder on HIP kernels, against the
reference-generated golden (resize) and the transformers-generated golden (encoder), plus their kernels vs torch fp32.
Stated tolerances: resize atol 2e-5 (fp32 both sides); encoder rel-L2 <= 3e-3 (fp16 MFMA operands, split-fp16 stream)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _g(s):
    return torch.Generator().manual_seed(s)


def test_resize_with_antialiasing_matches_reference_golden(golden_dir):
    from evoworld_amd.clip import resize_with_antialiasing
    g = np.load(f"{golden_dir}/resize_antialias.npz")
    y = resize_with_antialiasing(torch.tensor(g["x"]).to(DEV), (28, 28))
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], atol=2e-5)


def test_resize_full_size_vs_oracle_and_clip_normalisation():
    """576x1024 -> 224x224 (kernel sizes (3,7), sigma (0.786,1.786)) and the folded (x+1)/2, mean/std affine"""
    from evoworld_amd.clip import CLIP_MEAN, CLIP_STD, encode_image_preprocess
    from oracle.clip_ref import resize_with_antialiasing_ref
    img = torch.rand(1, 3, 576, 1024, generator=_g(1))
    want = (resize_with_antialiasing_ref(img * 2 - 1, (224, 224)) + 1) / 2
    want = (want - torch.tensor(CLIP_MEAN)[None, :, None, None]) / torch.tensor(CLIP_STD)[None, :, None, None]
    got = encode_image_preprocess(img.to(DEV))
    assert got.shape == (1, 3, 224, 224)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=3e-5)


@pytest.mark.parametrize("n_seq,S,heads,D", [(2, 257, 16, 80), (1, 17, 4, 80), (3, 64, 2, 64)])
def test_attn_small_vs_sdpa(n_seq, S, heads, D):
    from evoworld_amd import ops
    C = heads * D
    qkv = torch.randn(n_seq * S, 3 * C, generator=_g(2)).half().to(DEV)
    o = torch.empty(n_seq * S, C, dtype=torch.float16, device=DEV)
    ops.attn_small(qkv, qkv[:, C:], qkv[:, 2 * C:], o, n_seq, S, heads, D, 3 * C, C, D ** -0.5)
    q, k, v = (qkv[:, i * C:(i + 1) * C].float().reshape(n_seq, S, heads, D).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n_seq * S, C)
    assert rel_l2(o.float().cpu(), ref.cpu()) < 1e-3


def test_gemm_gelu_epilogue():
    from evoworld_amd import _lib, ops
    lib = _lib.load()
    for gen in (3, 2, 1):
        lib.ew_set_gemm_generation(gen)
        try:
            for (M, N, K) in ((257, 5120, 1280), (1100, 640, 320)):
                x = torch.randn(M, K, generator=_g(3)).half().to(DEV)
                w = (torch.randn(N, K, generator=_g(4)) / math.sqrt(K)).half().to(DEV)
                b = torch.randn(N, generator=_g(5)).half().to(DEV)
                out = ops.linear(x, w, b, act=ops.ACT_GELU)
                ref = F.gelu(x.float() @ w.float().T + b.float())
                assert rel_l2(out.float().cpu(), ref.cpu()) < 1e-3, (gen, M, N, K)
        finally:
            lib.ew_set_gemm_generation(3)


def test_clip_tiny_vs_transformers_golden(golden_dir):
    from evoworld_amd.clip import CLIPVisionModelWithProjection, DEFAULT_CLIP_CONFIG, random_clip_state_dict
    from oracle.clip_ref import tiny_clip_config
    g = np.load(f"{golden_dir}/clip_tiny.npz")
    cfg = tiny_clip_config()
    sd = {k: v.half().float() for k, v in random_clip_state_dict({**DEFAULT_CLIP_CONFIG, **cfg}, 0).items()}
    enc = CLIPVisionModelWithProjection(**cfg).load_state_dict(sd, device=DEV)
    y = enc(torch.tensor(g["x"]).to(DEV)).image_embeds
    e = rel_l2(y.cpu(), torch.tensor(g["image_embeds"]))
    print(f"CLIP tiny (head_dim 80) vs transformers golden rel-L2 {e:.3e}")
    assert y.shape == (2, 64) and e < 3e-3


def test_clip_vit_h_shape_runs_and_matches_oracle():
    """full ViT-H/14 geometry (1280 wide, 16 heads of 80, 257 tokens) with 2 layers: HIP vs the fp32 restatement"""
    from evoworld_amd.clip import CLIPVisionModelWithProjection, DEFAULT_CLIP_CONFIG, random_clip_state_dict
    from oracle.clip_ref import CLIPVisionRef
    cfg = dict(DEFAULT_CLIP_CONFIG, num_hidden_layers=2)
    sd = {k: v.half().float() for k, v in random_clip_state_dict(cfg, 1).items()}
    enc = CLIPVisionModelWithProjection(**cfg).load_state_dict(sd, device=DEV)
    x = torch.randn(1, 3, 224, 224, generator=_g(6))
    y = enc(x.to(DEV)).image_embeds
    want = CLIPVisionRef(**cfg).load_state_dict(sd)(x)
    e = rel_l2(y.cpu(), want)
    print(f"CLIP ViT-H geometry (2 layers) rel-L2 {e:.3e}")
    assert y.shape == (1, 1024) and e < 3e-3


def test_pipeline_encode_image_uses_reference_preprocessing():
    """pipeline with real vae / image_encoder components: embeddings = CLIP(preprocess(first frame)) (pipeline_evoworld.py:255-305)"""
    from evoworld_amd.clip import CLIPVisionModelWithProjection, DEFAULT_CLIP_CONFIG, encode_image_preprocess, random_clip_state_dict
    from evoworld_amd.pipeline import StableVideoDiffusionPipeline
    from evoworld_amd.unet import DEFAULT_CONFIG, UNetSpatioTemporalConditionModel, random_state_dict
    from oracle.unet_ref import tiny_config
    ccfg = dict(DEFAULT_CLIP_CONFIG, num_hidden_layers=1, projection_dim=64)
    enc = CLIPVisionModelWithProjection(**ccfg).load_state_dict(random_clip_state_dict(ccfg, 2), device=DEV)
    ucfg = tiny_config()
    unet = UNetSpatioTemporalConditionModel(**ucfg).load_state_dict(random_state_dict({**DEFAULT_CONFIG, **ucfg}, 0), device=DEV)
    pipe = StableVideoDiffusionPipeline(unet=unet, image_encoder=enc)
    img = torch.rand(1, 3, 128, 256, generator=_g(7)).to(DEV)
    got = pipe._encode_image(img)
    want = enc(encode_image_preprocess(img)).image_embeds.unsqueeze(1)
    assert got.shape == (1, 1, 64) and torch.equal(got, want)
