"""Calibration: this build's GEMM against torch.nn.functional.linear (rocBLAS / hipBLASLt assembly kernels) on the plain
(bias-only) shapes of the U-Net and on a few large-K shapes.  fp16 in, fp16 out, fp32 accumulate."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import ops
import tools.bench_kernels as B
for (M, N, K) in ((460800, 960, 320), (460800, 640, 320), (115200, 1920, 640), (28800, 3840, 1280), (28800, 1280, 5120),
                  (115200, 640, 2560), (460800, 320, 1280), (460800, 2560, 320), (28800, 10240, 1280), (8192, 8192, 8192)):
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    t_mine = B.timeit(lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b))
    t_lib = B.timeit(lambda: F.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: this build {t_mine:7.3f} ms {fl / t_mine / 1e9:7.0f} TF/s   torch/hipBLASLt {t_lib:7.3f} ms {fl / t_lib / 1e9:7.0f} TF/s", flush=True)
