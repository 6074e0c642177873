"""Whole-forward A/B of two library builds in one process (working tree vs libevoworld_hip_base.so): every ew_* entry point the
U-Net uses is swapped between the two libraries; alternating rounds, best-of."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import _lib
from evoworld_amd.unet import UNetSpatioTemporalConditionModel
new = _lib.load()
base = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libevoworld_hip_base.so"))
names = ("ew_gemm_f16", "ew_groupnorm_stats_f16", "ew_groupnorm_apply_f16", "ew_layernorm_f16", "ew_attn_spatial_f16", "ew_attn_temporal_f16")
fn_new = {n: getattr(new, n) for n in names}
fn_base = {}
for n in names:
    f = getattr(base, n); f.argtypes = fn_new[n].argtypes; f.restype = fn_new[n].restype; fn_base[n] = f
torch.manual_seed(0)
unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device="cuda")
B, T, h, w = 2, 25, 72, 128
x = torch.randn(B * T * h * w, 64, device="cuda", dtype=torch.float16); x[:, 18:] = 0
ehs = torch.randn(B, 1, 1024, device="cuda", dtype=torch.float16)
added = torch.tensor([[6.0, 127.0, 0.02]] * B, device="cuda")
def fwd(): return unet.forward_nhwc(x, 1.234, ehs, added, B, T, h, w)
def timeit(n=3):
    fwd(); torch.cuda.synchronize(); best = 1e9
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fwd(); e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e))
    return best
only = sys.argv[1:]          # optionally swap only these entry points
for rnd in range(3):
    for name, fns in (("base", fn_base), ("new", fn_new)):
        for n in names:
            setattr(new, n, fns[n] if (not only or n in only) else fn_new[n])
        print(f"{name}: {timeit():.2f} ms", flush=True)
for n in names: setattr(new, n, fn_new[n])
