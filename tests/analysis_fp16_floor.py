"""Analysis helper (not a test; lives under tests/ because only tests may import oracle/): how much of the HIP path's
rel-L2 distance to the fp32 oracle is the irreducible cost of fp16 MFMA operands, and how much is fp16 STORAGE of activations?
Runs the fp32 oracle U-Net (tiny config, CPU) three ways on the same inputs:
  (a) operands of every conv / linear rounded to fp16 (weights too), everything else fp32   -> floor of any fp16-MFMA design
  (b) (a) + every conv / linear / norm OUTPUT rounded to fp16                                -> what this build does
  (c) fp32 operands but convs with TF32-like 10-bit operand mantissas                        -> what the reference's own
      fp32 inference does on a GPU with PyTorch's default cudnn.allow_tf32=True
Usage: python tests/analysis_fp16_floor.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evoworld_amd.unet import DEFAULT_CONFIG, random_state_dict  # noqa: E402  (host-side weight generator only)
from oracle.unet_ref import UNetSpatioTemporalConditionModelRef as UNetRef, tiny_config  # noqa: E402


def rel(a, b):
    return float((a - b).norm() / b.norm())


def r16(t):
    return t.to(torch.float16).to(torch.float32)


def tf32(t):  # keep 10 explicit mantissa bits (round to nearest even on the fp32 bit pattern)
    i = t.contiguous().view(torch.int32)
    i = (i + 0x00000FFF + ((i >> 13) & 1)) & ~0x1FFF
    return i.view(torch.float32)


def run(model, inputs, pre=None, post=None, kinds_pre=(), kinds_post=()):
    hooks = []
    for m in model.modules():
        if pre and isinstance(m, kinds_pre):
            hooks.append(m.register_forward_pre_hook(lambda mod, args: tuple(pre(a) if torch.is_tensor(a) and a.is_floating_point() else a for a in args)))
        if post and isinstance(m, kinds_post):
            hooks.append(m.register_forward_hook(lambda mod, args, out: post(out) if torch.is_tensor(out) else out))
    try:
        with torch.no_grad():
            return model(*inputs)
    finally:
        for h in hooks:
            h.remove()


def main():
    torch.manual_seed(0)
    cfg = tiny_config()
    model = UNetRef(**cfg).eval()
    g = torch.Generator().manual_seed(1)
    model.load_state_dict({k: v.float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}, strict=True)
    T, h, w = cfg["num_frames"], 16, 32
    x = torch.randn(2, T, 18, h, w, generator=g)
    ehs = torch.randn(2, 1, cfg["cross_attention_dim"], generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    inputs = (x, torch.tensor(1.234), ehs, ids)
    mm = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)
    norms = (torch.nn.GroupNorm, torch.nn.LayerNorm)
    ref = run(model, inputs)
    w32 = {k: v.clone() for k, v in model.state_dict().items()}
    # (c) TF32-like conv operands, fp32 linears
    convs = (torch.nn.Conv2d, torch.nn.Conv3d)
    for m in model.modules():
        if isinstance(m, convs):
            m.weight.data = tf32(m.weight.data)
    c = run(model, inputs, pre=tf32, kinds_pre=convs)
    model.load_state_dict(w32)
    for m in model.modules():
        if isinstance(m, mm):
            m.weight.data = r16(m.weight.data)
            if m.bias is not None:
                m.bias.data = r16(m.bias.data)
    a = run(model, inputs, pre=r16, kinds_pre=mm)
    b = run(model, inputs, pre=r16, kinds_pre=mm, post=r16, kinds_post=mm + norms)
    print(f"(a) fp16 operands only            : rel-L2 vs fp32 = {rel(a, ref):.2e}")
    print(f"(b) fp16 operands + fp16 storage   : rel-L2 vs fp32 = {rel(b, ref):.2e}")
    print(f"(c) TF32 conv operands (reference) : rel-L2 vs fp32 = {rel(c, ref):.2e}")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def per_tensor_ablation():
    """Which residual-stream tensors need the split-fp16 (hi + lo) form?  Operands of every conv / linear are rounded to fp16
    (the floor), then ONE class of stream tensor at a time is additionally rounded to fp16 (oracle.unet_ref._rq tags); the
    added squared rel-L2 per class is what keeping that class in plain fp16 costs.  Result on the tiny config (x1e-6):
    resblock spatial out 0.25, resblock out 0.23, transformer out 0.19, proj_in 0.08, spatial attn 0.08, spatial ff 0.07,
    temporal ff_in 0.04, temporal attn 0.02 against a floor of 0.56 -> the last two are kept in plain fp16 by the product."""
    import oracle.unet_ref as O
    torch.manual_seed(0)
    cfg = tiny_config()
    model = UNetRef(**cfg).eval()
    g = torch.Generator().manual_seed(1)
    model.load_state_dict({k: v.half().float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}, strict=True)
    T, h, w = cfg["num_frames"], 16, 32
    x = torch.randn(2, T, 18, h, w, generator=g)
    ehs = torch.randn(2, 1, cfg["cross_attention_dim"], generator=g)
    ehs[0] = 0
    inputs = (x, torch.tensor(1.234), ehs, torch.tensor([[6.0, 127.0, 0.02]] * 2))
    mm = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)
    ref = run(model, inputs)
    base = rel(run(model, inputs, pre=r16, kinds_pre=mm), ref)
    print(f"operands only: {base:.3e}")
    for tag in ("res_sp", "res_out", "t_out", "proj_in", "s_attn", "s_ff", "t_ffin", "t_attn", "blend", "res_h1", "res_t1", "res_sc"):
        O.RES_Q = lambda t, tg, tag=tag: r16(t) if tg == tag else t
        e = rel(run(model, inputs, pre=r16, kinds_pre=mm), ref)
        print(f"  + fp16 {tag:8s}: {e:.3e}   added squared rel-L2 {(e * e - base * base) * 1e6:.3f}e-6")
    O.RES_Q = None


if __name__ == "__main__" and "--per-tensor" in sys.argv:
    per_tensor_ablation()


def per_timestep_floor(steps_list=(2, 3, 25)):
    """Round 3: the fp16-operand floor of WHOLE short clips.  For the inputs of tests/test_gpu_pipeline.py
    (tiny config, T=4, 16x32 latents, seed 5) the oracle denoise loop runs twice per step count: fp32 everywhere, and with the
    operands of every conv / linear (weights too) rounded to fp16 and everything else fp32 -- the floor of any design that feeds
    fp16 operands to the MFMA, before any storage or attention-internal rounding.  Prints, per step, the rel-L2 of the model
    output (eps) and of the latents, so the 2- / 3-step clip tolerances can be read against what fp16 operands alone cost at
    those noise levels."""
    from evoworld_amd.scheduler import EulerDiscreteScheduler
    from oracle.reproject_ref import euler_cfg_step_ref
    cfg = tiny_config()
    sd = {k: v.half().float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}
    model = UNetRef(**cfg).eval()
    model.load_state_dict(sd)
    mm = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)
    T, h, w = 4, 16, 32
    g = torch.Generator().manual_seed(5)
    lat0, il = torch.randn(1, T, 4, h, w, generator=g), torch.randn(1, T + 1, 4, h, w, generator=g)
    ehs, pl = torch.randn(1, 1, cfg["cross_attention_dim"], generator=g), torch.randn(1, T, 6, h, w, generator=g)
    il2 = torch.cat([torch.zeros_like(il), il])
    cond = torch.cat([il2[:, 0:1].repeat(1, T, 1, 1, 1), il2[:, 1:], torch.cat([pl, pl])], dim=2)
    e2 = torch.cat([torch.zeros_like(ehs), ehs])
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    guid = torch.linspace(1.0, 3.0, T)
    import oracle.unet_ref as O
    for attn_ops in (False, True):
        # attn_ops: ALSO the four matmul operands of the attention core (q, k, P, v) are fp16 -- every MFMA operand of the
        # network then is; nothing else (no storage rounding, fp32 softmax / norms / residual stream)
        O.ATTN_Q = r16 if attn_ops else None
        label = "fp16 operands of every conv / linear" + (" AND of the attention matmuls (q, k, P, v)" if attn_ops else "")
        for steps in steps_list:
            s = EulerDiscreteScheduler()
            s.set_timesteps(steps)
            lat_ref = lat0 * s.init_noise_sigma
            lat_q = lat_ref.clone()
            print(f"--- {steps}-step clip, {label} (sigmas {[round(float(v), 4) for v in s.sigmas[:4]]} ...)")
            for i in range(steps):
                sig, sign = float(s.sigmas[i]), float(s.sigmas[i + 1])
                xr = torch.cat([torch.cat([lat_ref, lat_ref]) / (sig ** 2 + 1) ** 0.5, cond], dim=2)
                xq = torch.cat([torch.cat([lat_q, lat_q]) / (sig ** 2 + 1) ** 0.5, cond], dim=2)
                O.ATTN_Q = None
                eps_r = run(model, (xr, s.timesteps[i], e2, ids))
                O.ATTN_Q = r16 if attn_ops else None
                eps_same = run(model, (xr, s.timesteps[i], e2, ids), pre=r16, kinds_pre=mm)        # same input: the per-forward floor
                eps_q = run(model, (xq, s.timesteps[i], e2, ids), pre=r16, kinds_pre=mm)           # own trajectory
                lat_ref = euler_cfg_step_ref(eps_r[0:1], eps_r[1:2], lat_ref, guid, sig, sign)
                lat_q = euler_cfg_step_ref(eps_q[0:1], eps_q[1:2], lat_q, guid, sig, sign)
                if steps <= 3 or i in (0, steps // 2, steps - 2, steps - 1):
                    print(f"  step {i:2d} sigma {sig:9.4f}: eps floor (same input) {rel(eps_same, eps_r):.2e}   latents after the step {rel(lat_q, lat_ref):.2e}")
            print(f"  => {steps}-step clip: final latents rel-L2 {rel(lat_q, lat_ref):.2e}")
    O.ATTN_Q = None


if __name__ == "__main__" and "--per-timestep" in sys.argv:
    per_timestep_floor()


def norm_input_ablation():
    """Round 3: what would it cost to feed the normalisation layers fp16-rounded inputs (i.e. to let GroupNorm statistics / apply or
    LayerNorm read the hi half of the split stream only and save its lo8 byte)?  On top of the fp16-operand floor (tiny config):
    GroupNorm inputs +0.42e-6 squared rel-L2 (7.49e-4 -> 9.92e-4: the whole remaining budget -- GroupNorm keeps reading hi + lo8, and
    ops.groupnorm(stats_hi_only=...) stays off), LayerNorm inputs +0.025e-6 (7.66e-4: affordable, worth ~0.7 ms; not taken)."""
    torch.manual_seed(0)
    cfg = tiny_config()
    model = UNetRef(**cfg).eval()
    g = torch.Generator().manual_seed(1)
    model.load_state_dict({k: v.half().float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}, strict=True)
    T, h, w = cfg["num_frames"], 16, 32
    x = torch.randn(2, T, 18, h, w, generator=g)
    ehs = torch.randn(2, 1, cfg["cross_attention_dim"], generator=g)
    ehs[0] = 0
    inputs = (x, torch.tensor(1.234), ehs, torch.tensor([[6.0, 127.0, 0.02]] * 2))
    mm = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)
    ref = run(model, inputs)
    base = rel(run(model, inputs, pre=r16, kinds_pre=mm), ref)
    print(f"operands only: {base:.3e}")
    for kinds, name in (((torch.nn.GroupNorm,), "GroupNorm inputs"), ((torch.nn.LayerNorm,), "LayerNorm inputs")):
        e = rel(run(model, inputs, pre=r16, kinds_pre=mm + kinds), ref)
        print(f"  + fp16 {name:18s}: {e:.3e}   added squared rel-L2 {(e * e - base * base) * 1e6:.3f}e-6")


if __name__ == "__main__" and "--norm-inputs" in sys.argv:
    norm_input_ablation()


def norm_stats_only_ablation():
    """Round 4: GroupNorm STATISTICS from the fp16-rounded input (the hi plane of the split stream alone), the
    apply still on the full-precision value.  Rounding noise is zero-mean over the >= 10^4 elements of a group, so only the
    statistics' own error matters -- measured added squared rel-L2 on top of the fp16-operand floor (tiny config): see the printout
    (the +0.42e-6 of --norm-inputs is the APPLY reading rounded values, not the statistics)."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    cfg = tiny_config()
    model = UNetRef(**cfg).eval()
    g = torch.Generator().manual_seed(1)
    model.load_state_dict({k: v.half().float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}, strict=True)
    T, h, w = cfg["num_frames"], 16, 32
    x = torch.randn(2, T, 18, h, w, generator=g)
    ehs = torch.randn(2, 1, cfg["cross_attention_dim"], generator=g)
    ehs[0] = 0
    inputs = (x, torch.tensor(1.234), ehs, torch.tensor([[6.0, 127.0, 0.02]] * 2))
    mm = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)
    ref = run(model, inputs)
    base = rel(run(model, inputs, pre=r16, kinds_pre=mm), ref)
    print(f"operands only: {base:.3e}")

    def gn_forward(self, t):
        n, c = t.shape[:2]
        tr = r16(t).reshape(n, self.num_groups, -1).double()
        mean = tr.mean(-1, keepdim=True)
        var = tr.var(-1, unbiased=False, keepdim=True)
        y = ((t.reshape(n, self.num_groups, -1).double() - mean) / torch.sqrt(var + self.eps)).float().reshape(t.shape)
        shape = (1, c) + (1,) * (t.ndim - 2)
        return y * self.weight.reshape(shape) + self.bias.reshape(shape)
    orig = torch.nn.GroupNorm.forward
    torch.nn.GroupNorm.forward = gn_forward
    try:
        e = rel(run(model, inputs, pre=r16, kinds_pre=mm), ref)
    finally:
        torch.nn.GroupNorm.forward = orig
    print(f"  + GroupNorm statistics from fp16-rounded inputs (apply on the full value): {e:.3e}   added squared rel-L2 {(e * e - base * base) * 1e6:.4f}e-6")


if __name__ == "__main__" and "--norm-stats" in sys.argv:
    norm_stats_only_ablation()


def per_group_energy():
    """Round 5: WHERE the fp16-operand distance comes from, per layer group, separately for the two operands.  fp32 oracle, tiny config; one
    group at a time gets (w) its weights rounded to fp16, everything else fp32, or (a) the inputs of its conv / linear modules rounded to fp16;
    the squared rel-L2 of the output is that group's share (the shares add up to the all-groups figure within 2 %: the terms are independent).
    Result (tiny config, shares of 0.55e-6 / 0.60e-6): conv_in 14.8 % / 12.2 %, conv_out 12.6 % / 7.9 %, level-0 proj_in + proj_out 12.9 % / 16.4 %,
    level-0 resblock convs 32.8 % / 32.9 %, everything below level 1 < 3 % -- i.e. ~40 % of the weight term sits in layers that are < 1 % of the
    flops.  The product splits exactly those operands (evoworld_amd/unet.py: split_operands)."""
    import collections
    cfg = tiny_config()
    sd = {k: v.float() for k, v in random_state_dict({**DEFAULT_CONFIG, **cfg}, 0).items()}
    g = torch.Generator().manual_seed(1)
    T, h, w = cfg["num_frames"], 16, 32
    inputs = (torch.randn(2, T, 18, h, w, generator=g), torch.tensor(1.234), torch.randn(2, 1, cfg["cross_attention_dim"], generator=g),
              torch.tensor([[6.0, 127.0, 0.02]] * 2))
    model = UNetRef(**cfg).eval()
    model.load_state_dict(sd, strict=True)
    mm = (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)

    def group_of(name):
        parts = name.split(".")
        top = parts[0] + ("." + parts[1] if parts[0] in ("down_blocks", "up_blocks") else "")
        if "attentions" in name:
            kind = "ff" if ".ff" in name else ("attn_proj" if ".attn" in name else ("pos_embed" if "time_pos_embed" in name else "proj_in/out"))
        elif "resnets" in name:
            kind = "res_temb" if "time_emb" in name else ("res_temporal" if "temporal_res_block" in name else "res_spatial")
        else:
            kind = "-"
        return top, kind

    groups = collections.defaultdict(list)
    for name, m in model.named_modules():
        if isinstance(m, mm):
            groups[group_of(name)].append((name, m))
    base = run(model, inputs)

    def e2(out):
        return float(((out - base).norm() / base.norm()) ** 2)

    def with_w(mods):
        keep = [(m, m.weight.data) for _, m in mods]
        for _, m in mods:
            m.weight.data = r16(m.weight.data)
        try:
            return e2(run(model, inputs))
        finally:
            for m, wd in keep:
                m.weight.data = wd

    def with_a(mods):
        hooks = [m.register_forward_pre_hook(lambda mod, args: tuple(r16(a) if torch.is_tensor(a) and a.is_floating_point() else a for a in args)) for _, m in mods]
        try:
            return e2(run(model, inputs))
        finally:
            for hk in hooks:
                hk.remove()

    every = [nm for ms in groups.values() for nm in ms]
    w_all, a_all = with_w(every), with_a(every)
    print(f"all weights rounded: {w_all:.3e} (rel-L2 {w_all ** 0.5:.3e}); all activation operands rounded: {a_all:.3e} (rel-L2 {a_all ** 0.5:.3e})")
    rows = [(gk, with_w(ms), with_a(ms)) for gk, ms in sorted(groups.items())]
    tw, ta = sum(r[1] for r in rows), sum(r[2] for r in rows)
    for gk, ew, ea in rows:
        if ew / tw > 0.004 or ea / ta > 0.004:
            print(f"  {gk[0]:16s} {gk[1]:12s} weights {ew:.3e} ({ew / tw * 100:5.1f} %)   activations {ea:.3e} ({ea / ta * 100:5.1f} %)")
    print(f"sum of groups: weights {tw:.3e}, activations {ta:.3e}")


if __name__ == "__main__" and "--per-group" in sys.argv:
    per_group_energy()
