"""Two full-size U-Net forwards (after one warm-up) for `rocprofv3 --kernel-trace`; attributes of the U-Net object can be set as
name=value arguments.  Usage: rocprofv3 --kernel-trace --output-format csv -d out -- python tools/prof_forward.py [attr=val ...]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd.unet import UNetSpatioTemporalConditionModel  # noqa: E402

unet = UNetSpatioTemporalConditionModel.from_random(seed=0, device="cuda")
for a in sys.argv[1:]:
    k, v = a.split("=")
    setattr(unet, k, type(getattr(unet, k))(int(v)))
B, T, h, w = 2, 25, 72, 128
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B * T * h * w, 64, device="cuda", dtype=torch.float16, generator=g)
x[:, 18:] = 0
ehs = torch.randn(B, 1, 1024, device="cuda", dtype=torch.float16, generator=g)
added = torch.tensor([[6.0, 127.0, 0.02]] * B, device="cuda")
n = int(os.environ.get("EW_PROF_FORWARDS", "3"))
for _ in range(n):
    unet.forward_nhwc(x, 1.234, ehs, added, B, T, h, w)
torch.cuda.synchronize()
