"""Epilogue cost of gen 3 per kernel class: normal vs epilogue skipped (debug bit 2) vs stores skipped (bit 1)."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib
import tools.bench_kernels as B
lib = _lib.load()
for rnd in range(2):
    for dbg, name in ((0, "full"), (1, "no stores"), (2, "no epilogue")):
        lib.ew_set_gemm_debug(dbg)
        print("##", name, flush=True)
        B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
        B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
        B.gemm_case("L0 qkv", 460800, 960, 320)
        B.gemm_case("L1 ff_down_res", 115200, 640, 2560, res=True)
        B.conv_case("L0 320", 50, 320, 320, 72, 128)
lib.ew_set_gemm_debug(0)
