"""LDS-DMA feed rate of the DMA-only ablation build for different tile shapes / ring depths (bytes staged per second per CU)."""
import ctypes, os, sys, math
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib, ops
import tools.bench_kernels as B
new = _lib.load()
prod = new.ew_gemm_f16
L = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libevoworld_hip_ab3.so"))
L.ew_gemm_f16.argtypes = prod.argtypes; L.ew_gemm_f16.restype = prod.restype
L.ew_set_gemm_debug.argtypes = [ctypes.c_int]
new.ew_gemm_f16 = L.ew_gemm_f16

def case(M, N, K, bm, bn, dbg, label):
    L.ew_set_gemm_debug(dbg)
    x, w, b = B.rnd(M, K), B.rnd(N, K) * 0.05, B.rnd(N)
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    ms = B.timeit(lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b))
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    byts = tiles * (K // 64) * (bm + bn) * 128
    print(f"{label:40s} M={M} N={N} K={K}: {ms:7.3f} ms  {byts / ms / 1e6 / 256:7.1f} GB/s per CU  ({byts / ms / 1e9:6.2f} TB/s chip)  tiles={tiles}", flush=True)

for rnd in range(2):
    case(28800, 1280, 5120, 256, 160, 0, "256x160 8w 3-stage (2 in flight)")
    case(28800, 1024, 5120, 128, 256, 0, "128x256 8w 3-stage")
    case(28800, 1024, 5120, 128, 128, 16, "128x128 4w x2 WG 2-stage")
    case(65536, 1280, 5120, 256, 160, 0, "256x160, M=65536 (1024 tiles/…)")
    case(65536, 1280, 1280, 256, 160, 0, "256x160, M=65536 K=1280")
    case(460800, 320, 1280, 256, 160, 0, "L0 ff_down 256x160")
L.ew_set_gemm_debug(0)
new.ew_gemm_f16 = prod
