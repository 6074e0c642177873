"""TEST INFRASTRUCTURE -- fp32 CPU oracle of the EvoWorld / SVD-Xtend spatio-temporal U-Net.

This file is a checker, not a product path: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  evoworld_amd/ never does.

PARITY UNPINNED: the arithmetic of these blocks lives in the third-party package
diffusers==0.31.0 (/root/reference/requirements.txt:36), which is absent from /root/reference and
from this image.  The reference holds no tests / golden vectors for it.  This is a
restatement of the published diffusers 0.31.0 architecture that the reference's U-Net shell
instantiates, anchored on the reference's own call sites:
  evoworld/trainer/unet_plucker.py:126-244   layer list, channel/heads config, conv_in/out, GN eps 1e-5
  evoworld/trainer/unet_plucker.py:355-488   forward orchestration (time embed, flatten, skip stack)
  evoworld/trainer/unet_plucker.py:7-13      which diffusers blocks are used
  evoworld/trainer/trainer_utils.py:17-64    in_channels = 4 + 4*n_cond + 4*n_memory + 6 = 18
  evoworld/trainer/train_evoworld.py:303-310 parameter-name substrings ("temporal_transformer_block",
                                             "conv_in", "conv_out", "norm") -> state-dict key layout
Module / parameter names follow the diffusers state-dict layout so a real
checkpoint's keys load 1:1.  `exec_dead_cross_attn=True` executes the single-KV-token cross attention
exactly as the reference does (q-proj, softmax over one logit, out-proj); the default computes the
algebraically identical out = to_out(to_v(ctx)).
"""
import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)` -> [cos | sin], fp32.
    Call sites: evoworld/trainer/unet_plucker.py:136,141."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    freqs = torch.exp(exponent)
    args = timesteps[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, time_embed_dim, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


ATTN_Q = None  # analysis knob (tests/analysis_fp16_floor.py): rounding applied to the attention core's matmul operands q, k, v, P
RES_Q = None   # analysis knob (tests/analysis_fp16_floor.py): a function applied at every residual-stream tensor


def _rq(x, tag=""):
    """residual-stream hook: RES_Q(x) or, when it accepts two arguments, RES_Q(x, tag)"""
    if RES_Q is None:
        return x
    try:
        return RES_Q(x, tag)
    except TypeError:
        return RES_Q(x)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = nn.GroupNorm(32, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = _rq(h + self.time_emb_proj(F.silu(temb))[:, :, None, None], "res_h1")    # analysis tag: GroupNorm input between the convs
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = _rq(self.conv_shortcut(x), "res_sc")
        return _rq(x + h, "res_sp")


class TemporalResnetBlock(nn.Module):
    def __init__(self, ch, temb_ch, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, ch, eps=eps)
        self.conv1 = nn.Conv3d(ch, ch, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_ch, ch)
        self.norm2 = nn.GroupNorm(32, ch, eps=eps)
        self.conv2 = nn.Conv3d(ch, ch, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x, temb):  # x [B,C,T,H,W], temb [B,T,Ct]
        h = self.conv1(F.silu(self.norm1(x)))
        t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None].permute(0, 2, 1, 3, 4)
        h = _rq(h + t, "res_t1")
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class AlphaBlender(nn.Module):
    """merge_strategy='learned_with_images'; image_only_indicator is all-zero on this path
    (evoworld/trainer/unet_plucker.py:428) so alpha = sigmoid(mix_factor) everywhere."""

    def __init__(self, alpha, switch):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([float(alpha)]))
        self.switch = switch

    def alpha(self):
        a = torch.sigmoid(self.mix_factor)
        return 1.0 - a if self.switch else a

    def forward(self, x_spatial, x_temporal):
        a = self.alpha().to(x_spatial.dtype)
        return a * x_spatial + (1.0 - a) * x_temporal


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, cin, cout, temb_ch, eps):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(cin, cout, temb_ch, eps)
        self.temporal_res_block = TemporalResnetBlock(cout, temb_ch, eps)
        # diffusers SpatioTemporalResBlock defaults switch_spatial_to_temporal_mix=False (only the TemporalVAE decoder
        # blocks pass True): out = a*spatial + (1-a)*temporal, a = sigmoid(mix_factor)
        self.time_mixer = AlphaBlender(0.5, switch=False)

    def forward(self, x, temb, T):
        x = self.spatial_res_block(x, temb)
        BF, C, H, W = x.shape
        B = BF // T
        xs = x.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)
        xt = self.temporal_res_block(xs, temb.reshape(B, T, -1))
        y = _rq(self.time_mixer(xs, xt), "res_out")
        return y.permute(0, 2, 1, 3, 4).reshape(BF, C, H, W)


class Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, cross_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim)])

    def forward(self, x, ctx=None, exec_dead=False):
        if ctx is not None and ctx.shape[1] == 1 and not exec_dead:
            # softmax over ONE key == 1  ->  out = to_out(to_v(ctx)), independent of q
            return self.to_out[0](self.to_v(ctx)).expand(x.shape[0], x.shape[1], -1)
        src = x if ctx is None else ctx
        q, k, v = self.to_q(x), self.to_k(src), self.to_v(src)
        B, S, _ = q.shape
        h = self.heads
        q = q.view(B, S, h, -1).transpose(1, 2)
        k = k.view(B, k.shape[1], h, -1).transpose(1, 2)
        v = v.view(B, v.shape[1], h, -1).transpose(1, 2)
        if ATTN_Q is None:
            o = F.scaled_dot_product_attention(q, k, v)
        else:       # analysis: the four MFMA operands of the attention core (q, k, P, v) rounded by ATTN_Q, everything else fp32
            q, k, v = ATTN_Q(q), ATTN_Q(k), ATTN_Q(v)
            p_ = torch.softmax(q @ k.transpose(-1, -2) * (q.shape[-1] ** -0.5), dim=-1)
            o = ATTN_Q(p_) @ v
        o = o.transpose(1, 2).reshape(B, S, -1)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.proj = nn.Linear(din, dout * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim_out or dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx, exec_dead=False):
        x = x + self.attn1(self.norm1(x))
        x = _rq(x + self.attn2(self.norm2(x), ctx, exec_dead), "s_attn")   # product: one epilogue (attn1 out-proj + folded attn2)
        x = _rq(x + self.ff(self.norm3(x)), "s_ff")
        return x


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm_in = nn.LayerNorm(dim, eps=1e-5)
        self.ff_in = FeedForward(dim, dim)
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, T, ctx, exec_dead=False):  # x [B*T, S, C]; ctx [B*S, 1, 1024]
        BF, S, C = x.shape
        B = BF // T
        x = x.reshape(B, T, S, C).permute(0, 2, 1, 3).reshape(B * S, T, C)
        x = _rq(x + self.ff_in(self.norm_in(x)), "t_ffin")
        x = x + self.attn1(self.norm1(x))
        x = _rq(x + self.attn2(self.norm2(x), ctx, exec_dead), "t_attn")
        x = x + self.ff(self.norm3(x))       # product: blended with the spatial branch in the same epilogue
        return x.reshape(B, S, T, C).permute(0, 2, 1, 3).reshape(BF, S, C)


class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, heads, dim_head, ch, cross_dim):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_dim)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, heads, dim_head, cross_dim)])
        self.time_pos_embed = TimestepEmbedding(ch, ch * 4, out_dim=ch)
        self.time_mixer = AlphaBlender(0.5, switch=False)
        self.proj_out = nn.Linear(inner, ch)
        self.ch = ch

    def forward(self, x, ehs, T, exec_dead=False):  # x [B*T,C,H,W]; ehs [B*T,1,1024]
        BF, C, H, W = x.shape
        B = BF // T
        S = H * W
        first = ehs.reshape(B, T, -1, ehs.shape[-1])[:, 0]                        # [B,1,1024]
        time_ctx = first[:, None].expand(B, S, first.shape[-2], first.shape[-1]).reshape(B * S, -1, first.shape[-1])
        res = x
        h = self.norm(x).permute(0, 2, 3, 1).reshape(BF, S, C)
        h = _rq(self.proj_in(h), "proj_in")
        frames = torch.arange(T, device=x.device).repeat(B)
        emb = self.time_pos_embed(timestep_embedding(frames, self.ch).to(h.dtype))[:, None, :]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            h = blk(h, ehs, exec_dead)
            hm = tblk(h + emb, T, time_ctx, exec_dead)     # product: h + emb is formed in the ff_in epilogue, never stored
            h = _rq(self.time_mixer(h, hm), "blend")
        h = self.proj_out(h)
        return _rq(h.reshape(BF, H, W, C).permute(0, 3, 1, 2) + res, "t_out")


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_ch, heads, cross_dim, layers, attn, downsample):
        super().__init__()
        eps = 1e-6 if attn else 1e-5
        self.resnets = nn.ModuleList(
            [SpatioTemporalResBlock(cin if i == 0 else cout, cout, temb_ch, eps) for i in range(layers)])
        self.attentions = nn.ModuleList(
            [TransformerSpatioTemporalModel(heads, cout // heads, cout, cross_dim) for _ in range(layers)]) if attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if downsample else None

    def forward(self, x, temb, ehs, T, exec_dead):
        outs = ()
        for i, r in enumerate(self.resnets):
            x = r(x, temb, T)
            if self.attentions is not None:
                x = self.attentions[i](x, ehs, T, exec_dead)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb_ch, heads, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(ch, ch, temb_ch, 1e-5) for _ in range(2)])
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(heads, ch // heads, ch, cross_dim)])

    def forward(self, x, temb, ehs, T, exec_dead):
        x = self.resnets[0](x, temb, T)
        x = self.attentions[0](x, ehs, T, exec_dead)
        return self.resnets[1](x, temb, T)


class UpBlock(nn.Module):
    def __init__(self, cin, prev, cout, temb_ch, heads, cross_dim, layers, attn, upsample):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(SpatioTemporalResBlock(rin + skip, cout, temb_ch, 1e-6))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList(
            [TransformerSpatioTemporalModel(heads, cout // heads, cout, cross_dim) for _ in range(layers)]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if upsample else None

    def forward(self, x, skips, temb, ehs, T, exec_dead):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips[-1]], dim=1)
            skips = skips[:-1]
            x = r(x, temb, T)
            if self.attentions is not None:
                x = self.attentions[i](x, ehs, T, exec_dead)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetSpatioTemporalConditionModelRef(nn.Module):
    """Defaults = evoworld/trainer/unet_plucker.py:69-94 with in_channels=18."""

    def __init__(self, in_channels=18, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 addition_time_embed_dim=256, projection_class_embeddings_input_dim=768,
                 layers_per_block=2, cross_attention_dim=1024, num_attention_heads=(5, 10, 20, 20),
                 num_frames=25):
        super().__init__()
        boc = tuple(block_out_channels)
        self.cfg = dict(in_channels=in_channels, out_channels=out_channels, block_out_channels=boc,
                        addition_time_embed_dim=addition_time_embed_dim,
                        projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
                        layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                        num_attention_heads=tuple(num_attention_heads), num_frames=num_frames)
        temb_ch = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_ch)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb_ch)
        n = len(boc)
        downs = []
        out = boc[0]
        for i in range(n):
            cin, out = out, boc[i]
            last = i == n - 1
            downs.append(DownBlock(cin, out, temb_ch, num_attention_heads[i], cross_attention_dim,
                                   layers_per_block, attn=not last, downsample=not last))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb_ch, num_attention_heads[-1], cross_attention_dim)
        rev, rheads = boc[::-1], tuple(num_attention_heads)[::-1]
        ups = []
        out = rev[0]
        for i in range(n):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, n - 1)]
            ups.append(UpBlock(cin, prev, out, temb_ch, rheads[i], cross_attention_dim,
                               layers_per_block + 1, attn=i > 0, upsample=i < n - 1))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, exec_dead_cross_attn=False,
                taps: Optional[dict] = None):
        """sample [B,T,Cin,h,w]; timestep 0-d tensor/float; ehs [B,1,1024]; added_time_ids [B,3]
        -> [B,T,4,h,w]   (evoworld/trainer/unet_plucker.py:355-488)"""
        B, T = sample.shape[:2]
        ts = torch.as_tensor(timestep, dtype=torch.float32, device=sample.device).reshape(-1).expand(B)
        c = self.cfg
        emb = self.time_embedding(timestep_embedding(ts, c["block_out_channels"][0]).to(sample.dtype))
        te = timestep_embedding(added_time_ids.flatten(), c["addition_time_embed_dim"]).reshape(B, -1).to(emb.dtype)
        emb = emb + self.add_embedding(te)
        x = sample.flatten(0, 1)
        emb = emb.repeat_interleave(T, dim=0)
        ehs = encoder_hidden_states.repeat_interleave(T, dim=0)
        x = self.conv_in(x)
        if taps is not None:
            taps["emb"] = emb
            taps["conv_in"] = x
        skips = (x,)
        for i, blk in enumerate(self.down_blocks):
            x, outs = blk(x, emb, ehs, T, exec_dead_cross_attn)
            skips += outs
            if taps is not None:
                taps[f"down{i}"] = x
        x = self.mid_block(x, emb, ehs, T, exec_dead_cross_attn)
        if taps is not None:
            taps["mid"] = x
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            x = blk(x, res, emb, ehs, T, exec_dead_cross_attn)
            if taps is not None:
                taps[f"up{i}"] = x
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return x.reshape(B, T, *x.shape[1:])


def tiny_config():
    """A shrunken config with the same topology (4 levels, same block types) for CPU-speed tests."""
    return dict(in_channels=18, out_channels=4, block_out_channels=(64, 128, 256, 256),
                addition_time_embed_dim=64, projection_class_embeddings_input_dim=192,
                layers_per_block=2, cross_attention_dim=64, num_attention_heads=(1, 2, 4, 4), num_frames=4)
