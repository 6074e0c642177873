"""Round 4: which Tensile solutions does hipBLASLt pick for the large-K shapes where it leads this build?
Run under `rocprofv3 --kernel-trace --stats`: the kernel names encode macro-tile (MT), DepthU, prefetch (PGR/PLR), LDS and wave-grid
parameters.  Study only -- nothing here is linked into the product."""
import torch
import torch.nn.functional as F
for (M, N, K) in ((28800, 1280, 5120), (28800, 10240, 1280), (115200, 640, 2560), (115200, 5120, 640)):
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.05
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    for _ in range(5):
        y = F.linear(x, w, b)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        y = F.linear(x, w, b)
    e.record(); torch.cuda.synchronize()
    t = a.elapsed_time(e) / 20
    print(f"M={M} N={N} K={K}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:.0f} TF/s", flush=True)
