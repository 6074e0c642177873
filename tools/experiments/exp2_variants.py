"""A/B of the GEMM workgroup shapes: variant 0 = 8 waves x 1 WG/CU, 3-stage ring; variant 1 (debug bit 16) = 4 waves x 2 WG/CU."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0]]
from evoworld_amd import _lib
import tools.bench_kernels as B
lib = _lib.load()
for rnd in range(1):
    for dbg in (0, 64):
        lib.ew_set_gemm_debug(dbg); print("variant dbg", dbg, "round", rnd)
        B.gemm_case("L0 qkv", 460800, 960, 320)
        B.gemm_case("L0 qk", 460800, 640, 320)
        B.gemm_case("L0 CxC res", 460800, 320, 320, res=True)
        B.gemm_case("L0 ff_down", 460800, 320, 1280, res=True)
        B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
        B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
        B.gemm_case("L2 ff_up_geglu", 28800, 10240, 1280, act=2)
        B.gemm_case("L2 ff_down", 28800, 1280, 5120, res=True)
        B.conv_case("L0 320", 50, 320, 320, 72, 128)
        B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
        B.convt_case("L0", 2, 25, 9216, 320)
