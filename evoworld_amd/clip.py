"""CLIP ViT image encoder with projection + the reference's antialiased 224x224 preprocessing, on the HIP kernels
(SURVEY.md §8f row N2).

Call surface of transformers.CLIPVisionModelWithProjection as evoworld/pipeline/pipeline_evoworld.py:255-305 drives it:
    enc = CLIPVisionModelWithProjection.from_pretrained(path, subfolder="image_encoder")       # or .from_random(seed)
    enc(pixel_values[N,3,224,224]).image_embeds -> [N, projection_dim]
plus `encode_image_preprocess(image[N,3,H,W] in [0,1])`: x*2-1 -> _resize_with_antialiasing((224,224)) -> (x+1)/2 -> CLIP
mean/std normalisation (pipeline_evoworld.py:264-285 with feature_extractor(do_normalize=True, do_resize=False, ...)).
State-dict keys are the HF keys (vision_model.embeddings.*, vision_model.encoder.layers.N.*, visual_projection.weight).

ViT-H/14 is 257 tokens per image and ONE image per clip: every contraction is ew_gemm_f16 (patch embedding as im2col GEMM,
fused q|k|v projection, MLP with the erf-GELU epilogue), LayerNorms are ew_layernorm_f16, the 16-head attention with
head_dim 80 runs on ew_attn_small_f16; the residual stream is split fp16 like the U-Net's."""
import json
import math
import os
from collections import OrderedDict
from types import SimpleNamespace

import torch

from . import ops
from .ops import ACT_GELU, Res

DEFAULT_CLIP_CONFIG = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16,
                           image_size=224, patch_size=14, projection_dim=1024, layer_norm_eps=1e-5, hidden_act="gelu")
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # transformers OPENAI_CLIP_MEAN / STD (CLIPImageProcessor defaults)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_param_spec(cfg):
    D, I, L, P = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"], cfg["patch_size"]
    n_pos = (cfg["image_size"] // P) ** 2 + 1
    spec = OrderedDict()
    e = "vision_model.embeddings."
    spec[e + "class_embedding"] = ((D,), 1.0)
    spec[e + "patch_embedding.weight"] = ((D, 3, P, P), 3 * P * P)
    spec[e + "position_embedding.weight"] = ((n_pos, D), 1.0)
    spec["vision_model.pre_layrnorm.weight"] = ((D,), "gamma"); spec["vision_model.pre_layrnorm.bias"] = ((D,), "beta")
    for i in range(L):
        p = f"vision_model.encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            spec[p + f"self_attn.{n}.weight"] = ((D, D), D); spec[p + f"self_attn.{n}.bias"] = ((D,), D)
        spec[p + "layer_norm1.weight"] = ((D,), "gamma"); spec[p + "layer_norm1.bias"] = ((D,), "beta")
        spec[p + "mlp.fc1.weight"] = ((I, D), D); spec[p + "mlp.fc1.bias"] = ((I,), D)
        spec[p + "mlp.fc2.weight"] = ((D, I), I); spec[p + "mlp.fc2.bias"] = ((D,), I)
        spec[p + "layer_norm2.weight"] = ((D,), "gamma"); spec[p + "layer_norm2.bias"] = ((D,), "beta")
    spec["vision_model.post_layernorm.weight"] = ((D,), "gamma"); spec["vision_model.post_layernorm.bias"] = ((D,), "beta")
    spec["visual_projection.weight"] = ((cfg["projection_dim"], D), D)
    return spec


def random_clip_state_dict(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, (shape, kind) in clip_param_spec(cfg).items():
        if kind == "gamma":
            sd[name] = 1.0 + 0.1 * (torch.rand(shape, generator=g) * 2 - 1)
        elif kind == "beta":
            sd[name] = 0.1 * (torch.rand(shape, generator=g) * 2 - 1)
        else:
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(kind)
    return sd


def resize_with_antialiasing(x, size=(224, 224), scale=None, shift=None):
    """pipeline_evoworld.py:746-774 on the device: Gaussian blur (sigma = max((factor-1)/2, 0.001), kernel = odd(max(4 sigma, 3)),
    reflect padding, x pass then y pass) followed by bicubic interpolation with align_corners=True.  x fp32 [N,C,H,W]."""
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sigmas[0], 3)), int(max(2.0 * 2 * sigmas[1], 3))]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]

    def gaussian(window, sigma):                     # `_gaussian` (:823-836), float32 on the host
        xs = torch.arange(window, dtype=torch.float32) - window // 2
        if window % 2 == 0:
            xs = xs + 0.5
        gk = torch.exp(-xs.pow(2.0) / (2 * torch.tensor(sigma, dtype=torch.float32).pow(2.0)))
        return (gk / gk.sum()).to(x.device).contiguous()
    x = x.contiguous()
    out = ops.blur_axis(x, gaussian(ks[1], sigmas[1]), axis=1)
    out = ops.blur_axis(out, gaussian(ks[0], sigmas[0]), axis=0)
    return ops.bicubic_resize(out, size[0], size[1], scale, shift)


def encode_image_preprocess(image01, image_mean=CLIP_MEAN, image_std=CLIP_STD):
    """image in [0,1] fp32 [N,3,H,W] -> CLIP pixel_values fp32 [N,3,224,224] (pipeline_evoworld.py:264-285)."""
    x = (image01.float() * 2.0 - 1.0).contiguous()
    dev = x.device
    mean, std = torch.tensor(image_mean, dtype=torch.float32), torch.tensor(image_std, dtype=torch.float32)
    # ((v + 1) / 2 - mean) / std = v * (0.5 / std) + (0.5 - mean) / std
    return resize_with_antialiasing(x, (224, 224), (0.5 / std).to(dev).contiguous(), ((0.5 - mean) / std).to(dev).contiguous())


class CLIPVisionModelWithProjection:
    def __init__(self, **config):
        cfg = dict(DEFAULT_CLIP_CONFIG)
        cfg.update({k: v for k, v in config.items() if k in DEFAULT_CLIP_CONFIG})
        if cfg["hidden_act"] != "gelu":
            raise NotImplementedError("only the erf GELU of CLIP ViT-H/14 (hidden_act='gelu') is built")
        if cfg["hidden_size"] % 64 or cfg["intermediate_size"] % 64 or (cfg["hidden_size"] // cfg["num_attention_heads"]) % 8:
            raise ValueError("evoworld_amd CLIP: hidden / intermediate sizes must be multiples of 64, head_dim of 8")
        self._cfg = cfg
        self.config = SimpleNamespace(**cfg)
        self.dtype = torch.float32
        self.device, self.w = None, None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, device="cuda", **_ignored):
        root = os.path.join(path, subfolder) if subfolder else path
        cfg = {}
        cj = os.path.join(root, "config.json")
        if os.path.exists(cj):
            raw = json.load(open(cj))
            raw = {**raw.get("vision_config", {}), **raw}          # CLIPVisionConfig fields may sit at the top level
            cfg = {k: raw[k] for k in DEFAULT_CLIP_CONFIG if k in raw}
        m = cls(**cfg)
        from safetensors.torch import load_file
        for fn in ("model.safetensors", "model.fp16.safetensors"):
            f = os.path.join(root, fn)
            if os.path.exists(f):
                return m.load_state_dict(load_file(f), device=device)
        raise FileNotFoundError(f"no model*.safetensors under {root}")

    @classmethod
    def from_random(cls, seed=0, device="cuda", **config):
        m = cls(**config)
        return m.load_state_dict(random_clip_state_dict(m._cfg, seed), device=device)

    def parameters(self):
        return iter([torch.empty(0, dtype=torch.float32)])        # `next(image_encoder.parameters()).dtype` (:262)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, _f=False):
        return self

    def load_state_dict(self, sd, device="cuda"):
        spec = clip_param_spec(self._cfg)
        missing = [k for k in spec if k not in sd]
        if missing:
            raise KeyError(f"CLIP state dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, (shape, _) in spec.items():
            if tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{k}: expected shape {shape}, got {tuple(sd[k].shape)}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("evoworld_amd.CLIPVisionModelWithProjection needs a GPU device (no CPU path)")
        dev = self.device
        cfg = self._cfg
        D, P = cfg["hidden_size"], cfg["patch_size"]

        def f32(k):
            return sd[k].to(device=dev, dtype=torch.float32)

        def h(t):
            return t.to(torch.float16).contiguous()
        W = {}
        kp = (3 * P * P + 63) // 64 * 64
        pe = torch.zeros(D, kp, device=dev)
        pe[:, : 3 * P * P] = f32("vision_model.embeddings.patch_embedding.weight").reshape(D, -1)
        W["patch"], W["kp"] = h(pe), kp
        W["cls"] = f32("vision_model.embeddings.class_embedding")
        W["pos"] = f32("vision_model.embeddings.position_embedding.weight")
        W["pre"] = (h(f32("vision_model.pre_layrnorm.weight")), h(f32("vision_model.pre_layrnorm.bias")))
        W["layers"] = []
        for i in range(cfg["num_hidden_layers"]):
            p = f"vision_model.encoder.layers.{i}."
            a = p + "self_attn."
            W["layers"].append({
                "ln1": (h(f32(p + "layer_norm1.weight")), h(f32(p + "layer_norm1.bias"))),
                "qkv": (h(torch.cat([f32(a + "q_proj.weight"), f32(a + "k_proj.weight"), f32(a + "v_proj.weight")])),
                        h(torch.cat([f32(a + "q_proj.bias"), f32(a + "k_proj.bias"), f32(a + "v_proj.bias")]))),
                "out": (h(f32(a + "out_proj.weight")), h(f32(a + "out_proj.bias"))),
                "ln2": (h(f32(p + "layer_norm2.weight")), h(f32(p + "layer_norm2.bias"))),
                "fc1": (h(f32(p + "mlp.fc1.weight")), h(f32(p + "mlp.fc1.bias"))),
                "fc2": (h(f32(p + "mlp.fc2.weight")), h(f32(p + "mlp.fc2.bias")))})
        W["post"] = (h(f32("vision_model.post_layernorm.weight")), h(f32("vision_model.post_layernorm.bias")))
        pd = cfg["projection_dim"]
        pw = torch.zeros((pd + 3) // 4 * 4, D, device=dev)
        pw[:pd] = f32("visual_projection.weight")
        W["proj"] = h(pw)
        self.w = W
        return self

    @torch.no_grad()
    def __call__(self, pixel_values, **_kw):
        if self.w is None:
            raise RuntimeError("weights not loaded")
        cfg, W = self._cfg, self.w
        S_img, P, D, H = cfg["image_size"], cfg["patch_size"], cfg["hidden_size"], cfg["num_attention_heads"]
        if pixel_values.ndim != 4 or tuple(pixel_values.shape[1:]) != (3, S_img, S_img):
            raise ValueError(f"pixel_values must be [N,3,{S_img},{S_img}], got {tuple(pixel_values.shape)}")
        x = pixel_values.to(device=self.device, dtype=torch.float32).contiguous()
        N = x.shape[0]
        G = (S_img // P) ** 2
        S = G + 1
        dev = x.device
        patches = ops.linear(ops.vit_patchify(x, P, W["kp"]), W["patch"])                  # [N*G, D]
        # token assembly ([class | patches] + position embedding): 257 x 1280 values per image -- host-side glue in fp32,
        # split into the (hi, lo8) stream
        tok = torch.cat([W["cls"].expand(N, 1, D), patches.float().reshape(N, G, D)], dim=1) + W["pos"][None]
        tok = tok.reshape(N * S, D)
        hs = Res.from_float(tok)
        h0 = ops.layernorm(hs, *W["pre"], eps=cfg["layer_norm_eps"])
        hs = Res(h0, None)                                       # pre_layrnorm output starts the stream
        hd = D // H
        for lw in W["layers"]:
            n1 = ops.layernorm(hs, *lw["ln1"], eps=cfg["layer_norm_eps"])
            qkv = ops.linear(n1, *lw["qkv"])
            ao = torch.empty(N * S, D, dtype=torch.float16, device=dev)
            ops.attn_small(qkv, qkv[:, D:], qkv[:, 2 * D:], ao, N, S, H, hd, 3 * D, D, hd ** -0.5)
            hs = ops.linear(ao, *lw["out"], out=Res.empty(N * S, D, dev, True), r1=hs, ld_r1=D)
            n2 = ops.layernorm(hs, *lw["ln2"], eps=cfg["layer_norm_eps"])
            f1 = ops.linear(n2, *lw["fc1"], act=ACT_GELU)
            hs = ops.linear(f1, *lw["fc2"], out=Res.empty(N * S, D, dev, True), r1=hs, ld_r1=D)
        last = hs.float().reshape(N, S, D)
        pooled_in = last[:, 0]                                    # class token
        pooled = ops.layernorm(Res.from_float(pooled_in), *W["post"], eps=cfg["layer_norm_eps"])
        emb = ops.linear(pooled, W["proj"])[:, : cfg["projection_dim"]].float()
        return SimpleNamespace(image_embeds=emb, last_hidden_state=last)
