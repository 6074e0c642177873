import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
import tools.bench_kernels as B
B.norm_cases()
import torch
from evoworld_amd import ops
for name, n, rows, C in (("gn3d L0", 2, 230400, 320), ("gn2d L1", 50, 2304, 640), ("gn2d L2", 50, 576, 1280), ("gn2d L3", 50, 144, 1280)):
    x, g, b = B.rnd(n * rows, C), B.rnd(C), B.rnd(C)
    out = torch.empty_like(x)
    sums = torch.zeros(n, 32, 2, dtype=torch.float32, device="cuda")
    ops.groupnorm_stats(x, sums, n, rows, C, 0, C)
    B.report(f"gn_apply only {name}", B.timeit(lambda: ops.groupnorm_apply(x, sums, g, b, out, n, rows, C, 0, C, 1e-5, True)), None, n * rows * C * 2 * 2)
    B.report(f"gn_stats only {name}", B.timeit(lambda: ops.groupnorm_stats(x, sums, n, rows, C, 0, C)), None, n * rows * C * 2)
