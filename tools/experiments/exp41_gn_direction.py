"""Round 3: does the GroupNorm apply pass find the tail of what the statistics pass just read in the Infinity Cache when it walks the tensor
in the opposite direction?  EW_GN_APPLY_REV=0|1 (read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from evoworld_amd import ops  # noqa: E402

for rows, C, slabs in ((9216, 320, 50), (2304, 640, 50), (9216, 640, 50)):
    g = torch.Generator().manual_seed(0)
    x = ops.Res.from_float(torch.randn(slabs * rows, C, generator=g).cuda())
    gam, bet = torch.ones(C, dtype=torch.float16, device="cuda"), torch.zeros(C, dtype=torch.float16, device="cuda")
    out = torch.empty(slabs * rows, C, dtype=torch.float16, device="cuda")
    pool = ops.WorkspacePool("cuda")
    fn = lambda: ops.groupnorm([x], gam, bet, slabs, rows, 1e-5, True, out=out, pool=pool)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 4)
    mb = slabs * rows * C * (3 + 3 + 2) / 1e6
    print(f"REV={os.environ.get('EW_GN_APPLY_REV', '0')} GroupNorm (stats + finalize + apply) {slabs}x{rows}x{C}: {best * 1e3:7.1f} us  {mb / best / 1e3:6.2f} TB/s", flush=True)
