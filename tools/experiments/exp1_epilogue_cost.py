import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv=[sys.argv[0]]
from evoworld_amd import ops, _lib
import tools.bench_kernels as B
lib=_lib.load()
for dbg in (0,1,2):
    lib.ew_set_gemm_debug(dbg); print("dbg",dbg)
    B.gemm_case("L0 qkv", 460800, 960, 320)
    B.gemm_case("L0 CxC res", 460800, 320, 320, res=True)
    B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
    B.gemm_case("L2 ff_up_geglu", 28800, 10240, 1280, act=2)
    B.conv_case("L0 320", 50, 320, 320, 72, 128)
