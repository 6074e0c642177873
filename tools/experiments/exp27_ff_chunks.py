"""Feed-forward pair (GEGLU up-projection -> down-projection + residual) issued over M chunks, so that the 4C-wide intermediate
of a chunk is still in the 256 MB Infinity Cache when the down-projection reads it.  Level 0/1 shapes; total time of the pair."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import ops
import tools.bench_kernels as B

def pair(M, C, nchunk):
    x, w1, b1 = B.rnd(M, C), B.rnd(8 * C, C) * 0.05, B.rnd(8 * C)
    w2, b2, r = B.rnd(C, 4 * C) * 0.05, B.rnd(C), B.rnd(M, C)
    mid = torch.empty(M, 4 * C, dtype=torch.float16, device="cuda")
    out = torch.empty(M, C, dtype=torch.float16, device="cuda")
    mc = M // nchunk
    def run():
        for i in range(nchunk):
            sl = slice(i * mc, (i + 1) * mc)
            ops.gemm(x[sl], w1, mid[sl], M=mc, N=8 * C, c1=C, lda=C, bias=b1, act=2)
            ops.gemm(mid[sl], w2, out[sl], M=mc, N=C, c1=4 * C, lda=4 * C, bias=b2, r1=r[sl], ld_r1=C)
    return B.timeit(run, iters=5, warm=2)

for rnd in range(2):
    for (M, C) in ((460800, 320), (115200, 640), (28800, 1280)):
        print(f"M={M} C={C}: " + "  ".join(f"{n} chunks {pair(M, C, n):6.3f} ms" for n in (1, 2, 4, 8, 16)), flush=True)
