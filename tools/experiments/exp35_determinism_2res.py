"""Is <0,23> (dense + row-bias + two residuals) deterministic run to run?  (exp34 reported different bits between two modes.)"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib, ops  # noqa: E402
import tools.bench_kernels as B  # noqa: E402

lib = _lib.load()
for (M, N, K) in ((115200, 640, 2560), (28800, 1280, 5120)):
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    r1, r2 = B.rnd(M, N), ops.Res.from_float(B.rnd(M, N).float())
    outs = []
    for i in range(6):
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, r1=r1, ld_r1=N, r2=r2, ld_r2=N, c_acc=0.5, c_r1=0.5, c_r2=0.5)
        torch.cuda.synchronize()
        outs.append(out)
    same = [torch.equal(outs[0], o) for o in outs[1:]]
    ref = 0.5 * (x.float() @ w.float().t() + b.float()) + 0.5 * r1.float() + 0.5 * r2.float()
    err = float((outs[0].float() - ref).norm() / ref.norm())
    print(f"{M}x{N}x{K} kernel {lib.ew_gemm_last_kernel().decode()}: run-to-run identical {same}, rel-L2 vs fp32 torch {err:.2e}, status {lib.ew_gemm_streamk_status()}", flush=True)
