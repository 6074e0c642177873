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
    lib.ew_set_gemm_generation.argtypes = [ctypes.c_int]
    lib.ew_set_gemm_generation(2)
    tot2, a2 = run(0)
    lib.ew_set_gemm_generation(3)
    tot3, a3 = run(0)
    lib.ew_set_gemm_generation(2)
    tot2b, a2b = run(0)
    lib.ew_set_gemm_generation(3)
    tot3b, a3b = run(0)
    print(f"forward (with per-call events): gen2 {tot2:.1f} / {tot2b:.1f} ms   gen3 {tot3:.1f} / {tot3b:.1f} ms")
    rows = sorted(a2.items(), key=lambda kv: -kv[1][1])
    print(f"{'call site':78s} {'n':>4s} {'gen2 ms':>8s} {'gen3 ms':>8s} {'gain':>7s} {'TF/s g3':>8s}")
    sg2 = sg3 = 0.0
    for key, (n, ms, fl) in rows:
        m2 = min(ms, a2b[key][1])
        m3 = min(a3[key][1], a3b[key][1])
        if key.startswith("gemm"):
            sg2 += m2
            sg3 += m3
        print(f"{key:78s} {n:4d} {m2:8.2f} {m3:8.2f} {100 * (m2 / m3 - 1):+6.1f}% {fl * n / m3 / 1e9 if fl else 0:8.0f}")
    print(f"GEMM total gen2 {sg2:.1f} ms, gen3 {sg3:.1f} ms")
finally:
    lib.ew_set_gemm_debug(0)
