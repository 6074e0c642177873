"""Round 5: the fused level-0 feed-forward with its MFMA accumulators in AccVGPRs (csrc/hipcc_agpr.sh: "amdgpu-agpr-alloc" set in the device IR).
Times the three call shapes of the forward (LayerNorm kernel NOT included) with the library EW_LIB_PATH points at; run once per library, alternating."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import ops  # noqa: E402
from evoworld_amd.ops import Res  # noqa: E402

M, C, S = 460800, 320, 18432
g = torch.Generator().manual_seed(0)
w1 = ((torch.rand(2560, C, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
b1 = ((torch.rand(2560, generator=g) * 2 - 1) / C ** 0.5).half().cuda()
w2 = ((torch.rand(C, 1280, generator=g) * 2 - 1) / 1280 ** 0.5).half().cuda()
b2 = ((torch.rand(C, generator=g) * 2 - 1) / 36).half().cuda()
h = Res.from_float(torch.randn(M, C, generator=g).cuda())
n3 = torch.randn(M, C, generator=g).half().cuda()
hp = torch.randn(M, C, generator=g).half().cuda()
pos = torch.randn(M // S, C, generator=g).half().cuda()
pack = ops.ff_pack(w1, b1, w2)
outR = Res.empty(M, C, "cuda", True)
outH = torch.empty(M, C, dtype=torch.float16, device="cuda")


def timeit(fn, name):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    print(f"{os.path.basename(os.environ.get('EW_LIB_PATH') or 'libevoworld_hip.so'):28s} {name:44s} {best * 1e3:8.1f} us", flush=True)


timeit(lambda: ops.ff_geglu320(n3, pack, b2, outR, r1=h), "spatial ff  (Res residual in / out)")
timeit(lambda: ops.ff_geglu320(n3, pack, b2, outH, r1=h, rowbias=pos, rows_per_group=S, ld_rowbias=C), "ff_in       (row-bias, fp16 out)")
timeit(lambda: ops.ff_geglu320(n3, pack, b2, outH, c_acc=0.5, r1=hp, c_r1=0.5, r2=h, c_r2=0.5), "temporal ff (AlphaBlender epilogue)")
