"""Does the row stride of the operands (L2 channel mapping) limit the LDS-DMA rate?  Dense GEMM M=28800 N=1280 with K=5120
(row stride 10240 B) against K=5248 (10496 B = 41 x 256) and K=5184 (10368 B = 81 x 128), full and DMA-only builds."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib
import tools.bench_kernels as B
new = _lib.load()
prod = new.ew_gemm_f16
L = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libevoworld_hip_ab3.so"))
L.ew_gemm_f16.argtypes = prod.argtypes; L.ew_gemm_f16.restype = prod.restype
for rnd in range(2):
    for name, fn in (("full", prod), ("DMA only", L.ew_gemm_f16)):
        new.ew_gemm_f16 = fn
        print("##", name, flush=True)
        for K in (5120, 5184, 5248, 4096, 4160):
            B.gemm_case(f"K={K}", 28800, 1280, K, res=True)
        for K in (1280, 1344):
            B.gemm_case(f"L0 ff_down K={K}", 460800, 320, K, res=True)
        for K in (640, 704):
            B.gemm_case(f"L1 geglu K={K}", 115200, 5120, K, act=2)
new.ew_gemm_f16 = prod
