"""Where does the U-Net forward go?  Times every ew_* C-ABI call of one full-size forward with HIP events (per call), grouped by
(symbol, shape key); GEMMs are additionally run with the epilogue compiled out at run time (debug bit 2) to split
main loop / epilogue.  In-process A/B only (box speed differs between gpurun calls).
Usage: python tools/experiments/exp10_forward_breakdown.py"""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import _lib, ops  # noqa: E402
from evoworld_amd.unet import UNetSpatioTemporalConditionModel  # noqa: E402

lib = _lib.load()
records = []          # (key, start_event, end_event)


def gemm_key(g):
    mode = {0: "dense", 1: "conv3x3", 2: "convT3"}[g.mode]
    epi = ("rb" if g.rowbias else "") + ("+r1" if g.r1 else "") + ("+r2" if g.r2 else "") + {0: "", 1: "+silu", 2: "+geglu"}[g.act]
    K = (g.c1 + g.c2) * {0: 1, 1: 9, 2: 3}[g.mode]
    extra = f" s{g.stride}u{g.upsample}" if g.mode == 1 else ""
    return f"gemm {mode:7s} M={g.M:6d} N={g.N:5d} K={K:5d}{extra} [{epi or 'bias'}]", 2.0 * g.M * g.N * K


def wrap(name):
    fn = getattr(lib, name)

    def timed(*a):
        if name == "ew_gemm_f16":
            key, fl = gemm_key(a[0]._obj)
        else:
            key, fl = name + " " + " ".join(str(x) for x in a if isinstance(x, int) and x < 10 ** 7)[:60], 0.0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a)
        e.record()
        records.append((key, fl, s, e))
        return r
    setattr(lib, name, timed)


for n in ("ew_gemm_f16", "ew_groupnorm_stats_f16", "ew_groupnorm_apply_f16", "ew_layernorm_f16", "ew_attn_spatial_f16",
          "ew_attn_temporal_f16"):
    wrap(n)

torch.manual_seed(0)
unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device="cuda")
B, T, h, w = 2, 25, 72, 128
x = torch.randn(B * T * h * w, 64, device="cuda", dtype=torch.float16)
x[:, 18:] = 0
ehs = torch.randn(B, 1, 1024, device="cuda", dtype=torch.float16)
added = torch.tensor([[6.0, 127.0, 0.02]] * B, device="cuda")


def forward():
    return unet.forward_nhwc(x, 1.234, ehs, added, B, T, h, w)


def run(dbg):
    lib.ew_set_gemm_debug(dbg)
    forward()
    torch.cuda.synchronize()
    records.clear()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    forward()
    e.record()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for key, fl, a, b in records:
        t = agg.setdefault(key, [0, 0.0, fl])
        t[0] += 1
        t[1] += a.elapsed_time(b)
    return s.elapsed_time(e), agg


try:
    sig = open(os.path.join(os.path.dirname(_lib.__file__), "unet.py")).read()
    tot0, a0 = run(0)
    tot2, a2 = run(2)
    tot0b, a0b = run(0)
    print(f"forward (with per-call events): {tot0:.1f} ms / {tot0b:.1f} ms; with GEMM epilogues skipped: {tot2:.1f} ms")
    rows = sorted(a0.items(), key=lambda kv: -kv[1][1])
    print(f"{'call site':78s} {'n':>4s} {'ms':>8s} {'ms(b)':>8s} {'no-epi':>8s} {'TF/s':>7s} {'TF/s no-epi':>11s}")
    sg = sn = 0.0
    for key, (n, ms, fl) in rows:
        msb = a0b[key][1]
        ms2 = a2.get(key, [0, 0.0])[1]
        tf = fl * n / min(ms, msb) / 1e9 if fl else 0
        tf2 = fl * n / ms2 / 1e9 if fl and ms2 else 0
        if key.startswith("gemm"):
            sg += min(ms, msb)
            sn += ms2
        print(f"{key:78s} {n:4d} {ms:8.2f} {msb:8.2f} {ms2:8.2f} {tf:7.0f} {tf2:11.0f}")
    print(f"GEMM total {sg:.1f} ms, with epilogue skipped {sn:.1f} ms")
finally:
    lib.ew_set_gemm_debug(0)
