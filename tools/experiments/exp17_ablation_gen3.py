"""Main-loop ablation on the compile-time ablation builds (make -C evoworld_amd/csrc ablate):
bit 1 = no MFMA, bit 2 = no ds_reads, bit 4 = no LDS-DMA.  'TF/s' is nominal (the flops of the full problem / time)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib  # noqa: E402
import tools.bench_kernels as B  # noqa: E402

new = _lib.load()
prod = new.ew_gemm_f16
d = os.path.dirname(_lib.__file__)
libs = [("full", prod)]
for a, what in ((4, "MFMA + ds_read (no DMA)"), (1, "ds_read + DMA (no MFMA)"), (3, "DMA only")):
    L = ctypes.CDLL(os.path.join(d, f"libevoworld_hip_g3ab{a}.so"))
    L.ew_gemm_f16.argtypes = prod.argtypes
    L.ew_gemm_f16.restype = prod.restype
    libs.append((what, L.ew_gemm_f16))
for rnd in range(2):
    for name, fn in libs:
        new.ew_gemm_f16 = fn
        print("##", name, flush=True)
        B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
        B.gemm_case("L2 ff_down_res", 28800, 1280, 5120, res=True)
        B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
new.ew_gemm_f16 = prod
