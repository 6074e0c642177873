"""AutoencoderKLTemporalDecoder on the HIP kernels: the temporal VAE either side of the denoise loop.

Call surface of the diffusers class the reference pipeline drives (evoworld/pipeline/pipeline_evoworld.py:307-328,358-385):
    vae = AutoencoderKLTemporalDecoder.from_pretrained(path, subfolder="vae")        # or .from_random(seed)
    vae.config.{block_out_channels, scaling_factor, force_upcast, latent_channels}; vae.dtype
    vae.encode(x[N,3,H,W] in [-1,1]).latent_dist.mode()   -> [N,4,H/8,W/8]
    vae.decode(z[N,4,h,w], num_frames=k).sample           -> [N,3,8h,8w]
State-dict keys are the diffusers keys (oracle/vae_ref.py lists the tree), so the SVD-XT VAE checkpoint loads.

Execution plan (same as the U-Net): fp16 channels-last activations, split-fp16 residual stream, every conv an
implicit-GEMM ew_gemm_f16 (3x3, stride-2 with the (0,1) padding of Downsample2D(padding=0) as `conv_shift`, nearest-x2
upsample as DMA addressing, frame-axis 3-tap), GroupNorm / SiLU as the deterministic two-stage kernels.  The single-head
512-wide attention of the mid blocks is two GEMMs around ew_softmax_rows_f16 per frame (scores leave the GEMM as hi + lo, so
the exponential sees fp32 scores); exact algebra removes work: the key bias shifts every score of a query row equally
(softmax-invariant, dropped), the value bias passes through the row-stochastic P unchanged (folded into the out-projection
bias), quant_conv (1x1, 8->8) is folded into the encoder's conv_out and only the 4 mean channels mode() returns are computed.
"""
import json
import math
import os
from collections import OrderedDict
from types import SimpleNamespace

import torch

from . import ops
from .ops import A_CONV3X3, A_CONVT3, Res

DEFAULT_VAE_CONFIG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                          scaling_factor=0.18215, force_upcast=True)
CPAD = 64


def vae_param_spec(cfg):
    """OrderedDict name -> (shape, fan_in or None) in module-registration order (diffusers key names)."""
    boc = tuple(cfg["block_out_channels"])
    L, lat = cfg["layers_per_block"], cfg["latent_channels"]
    spec = OrderedDict()

    def conv(p, o, i, k):
        spec[p + ".weight"] = ((o, i) + k, i * math.prod(k))
        spec[p + ".bias"] = ((o,), i * math.prod(k))

    def lin(p, o, i):
        spec[p + ".weight"] = ((o, i), i)
        spec[p + ".bias"] = ((o,), i)

    def norm(p, c):
        spec[p + ".weight"] = ((c,), "gamma")
        spec[p + ".bias"] = ((c,), "beta")

    def res2d(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, (3, 3)); norm(p + ".norm2", co); conv(p + ".conv2", co, co, (3, 3))
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, (1, 1))

    def st_res(p, ci, co):
        res2d(p + ".spatial_res_block", ci, co)
        t = p + ".temporal_res_block"
        norm(t + ".norm1", co); conv(t + ".conv1", co, co, (3, 1, 1)); norm(t + ".norm2", co); conv(t + ".conv2", co, co, (3, 1, 1))
        spec[p + ".time_mixer.mix_factor"] = ((1,), "mix")

    def attn(p, c):
        norm(p + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(p + "." + n, c, c)

    conv("encoder.conv_in", boc[0], 3, (3, 3))
    out = boc[0]
    for i, c in enumerate(boc):
        cin, out = out, c
        for l in range(L):
            res2d(f"encoder.down_blocks.{i}.resnets.{l}", cin if l == 0 else out, out)
        if i < len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", out, out, (3, 3))
    res2d("encoder.mid_block.resnets.0", boc[-1], boc[-1]); attn("encoder.mid_block.attentions.0", boc[-1])
    res2d("encoder.mid_block.resnets.1", boc[-1], boc[-1])
    norm("encoder.conv_norm_out", boc[-1]); conv("encoder.conv_out", 2 * lat, boc[-1], (3, 3))
    conv("decoder.conv_in", boc[-1], lat, (3, 3))
    st_res("decoder.mid_block.resnets.0", boc[-1], boc[-1]); attn("decoder.mid_block.attentions.0", boc[-1])
    st_res("decoder.mid_block.resnets.1", boc[-1], boc[-1])
    rev = boc[::-1]
    out = rev[0]
    for i, c in enumerate(rev):
        prev, out = out, c
        for l in range(L + 1):
            st_res(f"decoder.up_blocks.{i}.resnets.{l}", prev if l == 0 else out, out)
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", out, out, (3, 3))
    norm("decoder.conv_norm_out", boc[0]); conv("decoder.conv_out", 3, boc[0], (3, 3))
    conv("decoder.time_conv_out", 3, 3, (3, 1, 1))
    conv("quant_conv", 2 * lat, 2 * lat, (1, 1))
    return spec


def random_vae_state_dict(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, (shape, kind) in vae_param_spec(cfg).items():
        if kind == "gamma":
            sd[name] = torch.ones(shape)
        elif kind == "beta":
            sd[name] = torch.zeros(shape)
        elif kind == "mix":
            sd[name] = torch.full(shape, 0.3)
        else:
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(kind)
    return sd


class AutoencoderKLTemporalDecoder:
    def __init__(self, **config):
        cfg = dict(DEFAULT_VAE_CONFIG)
        cfg.update(config)
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        for c in cfg["block_out_channels"]:
            if c % 64:
                raise ValueError("evoworld_amd VAE: block_out_channels must be multiples of 64")
        self._cfg = cfg
        self.config = SimpleNamespace(**cfg)
        self.dtype = torch.float32          # API dtype: fp32 tensors in and out (the reference keeps the VAE in fp32, force_upcast)
        self.device, self.w = None, None
        self.split_residual = os.environ.get("EW_RESIDUAL", "split") != "fp16"
        self.chunk = 8                      # frames per encoder pass (the encoder has no frame-axis op: chunking is exact)

    # ---------------- construction ----------------
    @classmethod
    def from_pretrained(cls, path, subfolder=None, device="cuda", **_ignored):
        root = os.path.join(path, subfolder) if subfolder else path
        cfg = {}
        cj = os.path.join(root, "config.json")
        if os.path.exists(cj):
            cfg = {k: v for k, v in json.load(open(cj)).items() if k in DEFAULT_VAE_CONFIG}
        m = cls(**cfg)
        from safetensors.torch import load_file
        for fn in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"):
            f = os.path.join(root, fn)
            if os.path.exists(f):
                return m.load_state_dict(load_file(f), device=device)
        raise FileNotFoundError(f"no diffusion_pytorch_model*.safetensors under {root}")

    @classmethod
    def from_random(cls, seed=0, device="cuda", **config):
        m = cls(**config)
        return m.load_state_dict(random_vae_state_dict(m._cfg, seed), device=device)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, _f=False):
        return self

    def load_state_dict(self, sd, device="cuda"):
        spec = vae_param_spec(self._cfg)
        missing = [k for k in spec if k not in sd]
        if missing:
            raise KeyError(f"VAE state dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, (shape, _) in spec.items():
            if tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{k}: expected shape {shape}, got {tuple(sd[k].shape)}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("evoworld_amd.AutoencoderKLTemporalDecoder needs a GPU device (no CPU path)")
        self._pack(sd)
        return self

    # ---------------- weight packing ----------------
    def _pack(self, sd):
        dev = self.device
        boc, L, lat = self._cfg["block_out_channels"], self._cfg["layers_per_block"], self._cfg["latent_channels"]

        def f32(k):
            return sd[k].to(device=dev, dtype=torch.float32)

        def h(t):
            return t.to(torch.float16).contiguous()

        def conv(k, cpad=None):
            return ops.pack_conv_weight(f32(k + ".weight"), cpad), h(f32(k + ".bias"))

        def res2d(p, ci, co):
            d = {"n1": (h(f32(p + ".norm1.weight")), h(f32(p + ".norm1.bias"))), "c1": conv(p + ".conv1"),
                 "n2": (h(f32(p + ".norm2.weight")), h(f32(p + ".norm2.bias"))), "c2": conv(p + ".conv2"), "ci": ci, "co": co}
            if ci != co:
                d["sc"] = (h(f32(p + ".conv_shortcut.weight")[:, :, 0, 0]), h(f32(p + ".conv_shortcut.bias")))
            return d

        def st_res(p, ci, co):
            d = res2d(p + ".spatial_res_block", ci, co)
            t = p + ".temporal_res_block"
            d["tn1"] = (h(f32(t + ".norm1.weight")), h(f32(t + ".norm1.bias"))); d["t1"] = conv(t + ".conv1")
            d["tn2"] = (h(f32(t + ".norm2.weight")), h(f32(t + ".norm2.bias"))); d["t2"] = conv(t + ".conv2")
            d["mix"] = float(torch.sigmoid(f32(p + ".time_mixer.mix_factor")).item())
            return d

        def attn(p, c):
            wo, bo = f32(p + ".to_out.0.weight"), f32(p + ".to_out.0.bias")
            return {"gn": (h(f32(p + ".group_norm.weight")), h(f32(p + ".group_norm.bias"))),
                    "q": (h(f32(p + ".to_q.weight")), h(f32(p + ".to_q.bias"))),
                    "k": h(f32(p + ".to_k.weight")),                       # key bias: constant per query row -> softmax-invariant
                    "v": h(f32(p + ".to_v.weight")),                       # value bias: rows of P sum to 1 -> moves to the out bias
                    "o": (h(wo), h(wo @ f32(p + ".to_v.bias") + bo)), "c": c}

        W = {}
        W["e_in"] = conv("encoder.conv_in", CPAD)
        W["e_down"] = []
        out = boc[0]
        for i, c in enumerate(boc):
            cin, out = out, c
            blk = {"res": [res2d(f"encoder.down_blocks.{i}.resnets.{l}", cin if l == 0 else out, out) for l in range(L)]}
            if i < len(boc) - 1:
                blk["down"] = conv(f"encoder.down_blocks.{i}.downsamplers.0.conv")
            W["e_down"].append(blk)
        W["e_mid"] = (res2d("encoder.mid_block.resnets.0", boc[-1], boc[-1]), attn("encoder.mid_block.attentions.0", boc[-1]),
                      res2d("encoder.mid_block.resnets.1", boc[-1], boc[-1]))
        W["e_no"] = (h(f32("encoder.conv_norm_out.weight")), h(f32("encoder.conv_norm_out.bias")))
        # quant_conv (1x1) folded into conv_out; only the `lat` mean channels
        wq, bq = f32("quant_conv.weight")[:lat, :, 0, 0], f32("quant_conv.bias")[:lat]
        wo, bo = f32("encoder.conv_out.weight"), f32("encoder.conv_out.bias")
        W["e_out"] = (ops.pack_conv_weight(torch.einsum("qo,oikl->qikl", wq, wo)), h(wq @ bo + bq))
        W["d_in"] = conv("decoder.conv_in", CPAD)
        W["d_mid"] = (st_res("decoder.mid_block.resnets.0", boc[-1], boc[-1]), attn("decoder.mid_block.attentions.0", boc[-1]),
                      st_res("decoder.mid_block.resnets.1", boc[-1], boc[-1]))
        W["d_up"] = []
        rev = boc[::-1]
        out = rev[0]
        for i, c in enumerate(rev):
            prev, out = out, c
            blk = {"res": [st_res(f"decoder.up_blocks.{i}.resnets.{l}", prev if l == 0 else out, out) for l in range(L + 1)]}
            if i < len(rev) - 1:
                blk["up"] = conv(f"decoder.up_blocks.{i}.upsamplers.0.conv")
            W["d_up"].append(blk)
        W["d_no"] = (h(f32("decoder.conv_norm_out.weight")), h(f32("decoder.conv_norm_out.bias")))
        wc, bc = f32("decoder.conv_out.weight"), f32("decoder.conv_out.bias")
        W["d_out"] = (ops.pack_conv_weight(torch.cat([wc, torch.zeros_like(wc[:1])])), h(torch.cat([bc, torch.zeros(1, device=dev)])))  # 3 -> 4 rows
        W["d_time"] = (f32("decoder.time_conv_out.weight")[:, :, :, 0, 0].contiguous(), f32("decoder.time_conv_out.bias").contiguous())
        self.w = W

    # ---------------- building blocks (activations: fp16 [N*H*W, C], residual stream: Res) ----------------
    def _res(self, rows, C, dev):
        return Res.empty(rows, C, dev, self.split_residual)

    def _conv(self, x, wb, N, H, W_, Ho, Wo, res_out=False, **kw):
        w, b = wb
        M = N * Ho * Wo
        out = self._res(M, w.shape[0], x.device) if res_out else torch.empty(M, w.shape[0], dtype=torch.float16, device=x.device)
        return ops.gemm(x, w, out, M=M, N=w.shape[0], c1=x.shape[-1], lda=x.shape[-1], bias=b, mode=A_CONV3X3,
                        conv=(N, H, W_, Ho, Wo, kw.pop("stride", 1), kw.pop("upsample", 0)), **kw)

    def _convt(self, x, wb, B, T, P, res_out=False, **kw):
        w, b = wb
        out = self._res(B * T * P, w.shape[0], x.device) if res_out else torch.empty(B * T * P, w.shape[0], dtype=torch.float16, device=x.device)
        return ops.gemm(x, w, out, M=B * T * P, N=w.shape[0], c1=x.shape[-1], lda=x.shape[-1], bias=b, mode=A_CONVT3,
                        tconv=(B, T, P), **kw)

    def _resnet2d(self, d, x, N, H, W_):
        """ResnetBlock2D without time embedding: x + conv2(silu(gn(conv1(silu(gn(x))))))  (x: Res) -> Res"""
        HW = H * W_
        dev = x.hi.device
        hN = ops.groupnorm([x], *d["n1"], N, HW, 1e-6, True, pool=self._pool)
        h1 = self._conv(hN, d["c1"], N, H, W_, H, W_)
        h2 = ops.groupnorm([h1], *d["n2"], N, HW, 1e-6, True, pool=self._pool)
        if "sc" in d:
            sc = self._res(N * HW, d["co"], dev)
            ops.gemm(x.hi, d["sc"][0], sc, M=N * HW, N=d["co"], c1=d["ci"], lda=d["ci"], bias=d["sc"][1])
        else:
            sc = x
        return self._conv(h2, d["c2"], N, H, W_, H, W_, res_out=True, r1=sc, ld_r1=d["co"])

    def _st_resblock(self, d, x, B, T, H, W_):
        """SpatioTemporalResBlock of the decoder: AlphaBlender 'learned' with switch_spatial_to_temporal_mix=True:
        out = (1-a)*xsp + a*(xsp + conv2(..)) = xsp + a*conv2(..), a = sigmoid(mix_factor)."""
        HW = H * W_
        xsp = self._resnet2d(d, x, B * T, H, W_)
        g1 = ops.groupnorm([xsp], *d["tn1"], B, T * HW, 1e-5, True, pool=self._pool)
        t1 = self._convt(g1, d["t1"], B, T, HW)
        g2 = ops.groupnorm([t1], *d["tn2"], B, T * HW, 1e-5, True, pool=self._pool)
        return self._convt(g2, d["t2"], B, T, HW, res_out=True, r1=xsp, ld_r1=d["co"], c_acc=d["mix"], c_r1=1.0)

    def _attention(self, d, x, N, H, W_):
        """single-head attention over the H*W tokens of each frame, residual outside (diffusers Attention)"""
        C, S = d["c"], H * W_
        dev = x.hi.device
        if S % 64:
            raise ValueError("VAE attention needs (H/8)*(W/8) to be a multiple of 64")
        hn = ops.groupnorm([x], *d["gn"], N, S, 1e-6, False, pool=self._pool)
        q = ops.linear(hn, *d["q"])
        k = ops.linear(hn, d["k"])
        o = torch.empty(N * S, C, dtype=torch.float16, device=dev)
        vt = torch.empty(C, S, dtype=torch.float16, device=dev)
        scores = Res.empty(S, S, dev, True)                                 # hi + lo: the exponential sees fp32 scores
        p = torch.empty(S, S, dtype=torch.float16, device=dev)
        for f in range(N):
            sl = slice(f * S, (f + 1) * S)
            ops.gemm(d["v"], hn[sl], vt, M=C, N=S, c1=C, lda=C)             # V^T = W_v X^T (swapped operands), this frame
            ops.gemm(q[sl], k[sl], scores, M=S, N=S, c1=C, lda=C, c_acc=1.0 / math.sqrt(C))
            ops.softmax_rows(scores, out=p)
            ops.gemm(p, vt, o[sl], M=S, N=C, c1=S, lda=S)
        return ops.linear(o, *d["o"], out=self._res(N * S, C, dev), r1=x, ld_r1=C)

    def _begin(self, dev):
        if getattr(self, "_pool", None) is None or self._pool.buf.device != dev:
            self._pool = ops.WorkspacePool(dev, floats=16 << 20)
        self._pool.reset()

    # ---------------- encode ----------------
    def _encode_chunk(self, x):
        """x fp32 [n,3,H,W] -> fp32 [n,latent,H/8,W/8] (mean of the latent distribution)"""
        Wt = self.w
        n, _, H, W_ = x.shape
        dev = x.device
        self._begin(dev)
        xin = torch.zeros(n * H * W_, CPAD, dtype=torch.float16, device=dev)
        ops.nchw_f32_to_nhwc_f16(x.contiguous(), xin, CPAD)
        h = self._conv(xin, Wt["e_in"], n, H, W_, H, W_, res_out=True)
        for blk in Wt["e_down"]:
            for d in blk["res"]:
                h = self._resnet2d(d, h, n, H, W_)
            if "down" in blk:
                h = self._conv(h.hi, blk["down"], n, H, W_, H // 2, W_ // 2, res_out=True, stride=2, conv_shift=1)
                H, W_ = H // 2, W_ // 2
        r0, at, r1 = Wt["e_mid"]
        h = self._resnet2d(r0, h, n, H, W_)
        h = self._attention(at, h, n, H, W_)
        h = self._resnet2d(r1, h, n, H, W_)
        hn = ops.groupnorm([h], *Wt["e_no"], n, H * W_, 1e-6, True, pool=self._pool)
        lat = self._cfg["latent_channels"]
        z = self._conv(hn, Wt["e_out"], n, H, W_, H, W_)
        return ops.nhwc_f16_to_nchw_f32(z, n, lat, H, W_, lat)

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        if self.w is None:
            raise RuntimeError("weights not loaded")
        if x.ndim != 4 or x.shape[1] != 3 or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"encode expects [N,3,H,W] with H, W multiples of 8, got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=torch.float32)
        mean = torch.cat([self._encode_chunk(x[i:i + self.chunk]) for i in range(0, x.shape[0], self.chunk)], dim=0)

        def _no_sample(*_a, **_k):
            raise NotImplementedError("only latent_dist.mode() (what the pipeline uses, pipeline_evoworld.py:311) is computed")
        dist = SimpleNamespace(mode=lambda: mean, mean=mean, sample=_no_sample)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    # ---------------- decode ----------------
    @torch.no_grad()
    def decode(self, z, num_frames=1, return_dict=True):
        """z fp32 [N,4,h,w] (already divided by scaling_factor by the caller, pipeline_evoworld.py:360); N % num_frames == 0."""
        if self.w is None:
            raise RuntimeError("weights not loaded")
        if z.ndim != 4 or z.shape[1] != self._cfg["latent_channels"] or z.shape[0] % num_frames:
            raise ValueError(f"decode expects [N,{self._cfg['latent_channels']},h,w] with N a multiple of num_frames, got {tuple(z.shape)}")
        Wt = self.w
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        N, _, H, W_ = z.shape
        B, T = N // num_frames, num_frames
        dev = z.device
        self._begin(dev)
        zin = torch.zeros(N * H * W_, CPAD, dtype=torch.float16, device=dev)
        ops.nchw_f32_to_nhwc_f16(z, zin, CPAD)
        h = self._conv(zin, Wt["d_in"], N, H, W_, H, W_, res_out=True)
        r0, at, r1 = Wt["d_mid"]
        h = self._st_resblock(r0, h, B, T, H, W_)
        h = self._attention(at, h, N, H, W_)
        h = self._st_resblock(r1, h, B, T, H, W_)
        for blk in Wt["d_up"]:
            for d in blk["res"]:
                h = self._st_resblock(d, h, B, T, H, W_)
            if "up" in blk:
                h = self._conv(h.hi, blk["up"], N, H, W_, 2 * H, 2 * W_, res_out=True, upsample=1)
                H, W_ = 2 * H, 2 * W_
        hn = ops.groupnorm([h], *Wt["d_no"], N, H * W_, 1e-6, True, pool=self._pool)
        y = self._conv(hn, Wt["d_out"], N, H, W_, H, W_)                               # [N*H*W, 4] (3 real channels)
        img = ops.nhwc_f16_to_nchw_f32(y, N, 3, H, W_, 4).reshape(B, T, 3, H, W_)
        out = ops.time_conv3(img, *Wt["d_time"]).reshape(N, 3, H, W_)
        return SimpleNamespace(sample=out) if return_dict else (out,)
