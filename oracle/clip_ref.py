"""ORACLE (test infrastructure): fp32 PyTorch restatement of transformers' CLIPVisionModelWithProjection (3P; the pipeline's
image_encoder, evoworld/pipeline/pipeline_evoworld.py:255-305) and of the reference's `_resize_with_antialiasing`
(:746-850).  Pinned: tests/golden/clip_tiny.npz holds the output of the REAL transformers implementation installed in the
build container (oracle/make_goldens_clip.py) for seeded weights; resize_antialias.npz pins the preprocessing."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def resize_with_antialiasing_ref(x, size):
    """pipeline_evoworld.py:746-850 restated (same float32 op order)"""
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sigmas[0], 3)), int(max(2.0 * 2 * sigmas[1], 3))]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]

    def gaussian(window, sigma):
        xs = torch.arange(window, dtype=torch.float32) - window // 2
        if window % 2 == 0:
            xs = xs + 0.5
        g = torch.exp(-xs.pow(2.0) / (2 * torch.tensor(sigma, dtype=torch.float32).pow(2.0)))
        return g / g.sum()

    def filt(t, kern, axis):
        k = kern.numel()
        pf = (k - 1) // 2
        pad = (pf, k - 1 - pf, 0, 0) if axis == 1 else (0, 0, pf, k - 1 - pf)
        t = F.pad(t, pad, mode="reflect")
        c = t.shape[1]
        wgt = (kern.reshape(1, 1, 1, k) if axis == 1 else kern.reshape(1, 1, k, 1)).expand(c, 1, -1, -1)
        return F.conv2d(t, wgt, groups=c)
    x = filt(x, gaussian(ks[1], sigmas[1]), 1)
    x = filt(x, gaussian(ks[0], sigmas[0]), 0)
    return F.interpolate(x, size=size, mode="bicubic", align_corners=True)


class CLIPVisionRef(nn.Module):
    def __init__(self, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                 patch_size=14, projection_dim=1024, layer_norm_eps=1e-5, **_):
        super().__init__()
        self.cfg = dict(D=hidden_size, H=num_attention_heads, P=patch_size, eps=layer_norm_eps, L=num_hidden_layers)
        self.sd = None

    def load_state_dict(self, sd, strict=True):
        self.sd = {k: v.float() for k, v in sd.items()}
        return self

    @torch.no_grad()
    def forward(self, pixel_values):
        sd, c = self.sd, self.cfg
        D, H, P, eps = c["D"], c["H"], c["P"], c["eps"]
        x = F.conv2d(pixel_values, sd["vision_model.embeddings.patch_embedding.weight"], stride=P).flatten(2).transpose(1, 2)
        N = x.shape[0]
        x = torch.cat([sd["vision_model.embeddings.class_embedding"].expand(N, 1, D), x], dim=1)
        x = x + sd["vision_model.embeddings.position_embedding.weight"][None]
        x = F.layer_norm(x, (D,), sd["vision_model.pre_layrnorm.weight"], sd["vision_model.pre_layrnorm.bias"], eps)
        for i in range(c["L"]):
            p = f"vision_model.encoder.layers.{i}."
            h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
            q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * (D // H) ** -0.5
            k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
            v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
            S = x.shape[1]
            q, k, v = (t.reshape(N, S, H, D // H).transpose(1, 2) for t in (q, k, v))
            a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
            a = a.transpose(1, 2).reshape(N, S, D)
            x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
            h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
            h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
            x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        pooled = F.layer_norm(x[:, 0], (D,), sd["vision_model.post_layernorm.weight"], sd["vision_model.post_layernorm.bias"], eps)
        return F.linear(pooled, sd["visual_projection.weight"])


def tiny_clip_config():
    """head_dim 80 like ViT-H/14 (1280 / 16), 17 tokens"""
    return dict(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14,
                projection_dim=64, layer_norm_eps=1e-5, hidden_act="gelu")
