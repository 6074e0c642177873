"""Whole-forward A/B of runtime switches in ONE process (full-size U-Net, B=2, T=25, 72x128 latents): alternating rounds,
best-of-3 per round.  Usage: python tools/ab_forward_flags.py name=attr:val[,attr:val] ...   (attributes of the U-Net object, or
dbg:<bits> for ew_set_gemm_debug).  Example: python tools/ab_forward_flags.py base=fused_gn_stats:0 fused=fused_gn_stats:1"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import _lib  # noqa: E402
from evoworld_amd.unet import UNetSpatioTemporalConditionModel  # noqa: E402

lib = _lib.load()
variants = []
for a in sys.argv[1:]:
    name, _, spec = a.partition("=")
    variants.append((name, [kv.split(":") for kv in spec.split(",") if kv]))
unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device="cuda")
B, T, h, w = 2, 25, 72, 128
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B * T * h * w, 64, device="cuda", dtype=torch.float16, generator=g)
x[:, 18:] = 0
ehs = torch.randn(B, 1, 1024, device="cuda", dtype=torch.float16, generator=g)
added = torch.tensor([[6.0, 127.0, 0.02]] * B, device="cuda")


def apply(kvs):
    lib.ew_set_gemm_debug(0)
    for k, v in kvs:
        if k == "dbg":
            lib.ew_set_gemm_debug(int(v))
        else:
            setattr(unet, k, type(getattr(unet, k))(int(v)))


def fwd():
    return unet.forward_nhwc(x, 1.234, ehs, added, B, T, h, w)


def timeit(n=3):
    fwd()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fwd()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best, out


outs = {}
for rnd in range(3):
    for name, kvs in variants:
        apply(kvs)
        ms, out = timeit()
        outs.setdefault(name, out.float().clone())
        print(f"round {rnd} {name:12s} {ms:8.2f} ms", flush=True)
names = [n for n, _ in variants]
for n in names[1:]:
    d = (outs[n] - outs[names[0]]).norm() / outs[names[0]].norm()
    print(f"rel-L2 {n} vs {names[0]}: {float(d):.3e}")
lib.ew_set_gemm_debug(0)
