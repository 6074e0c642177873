import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0]]
from evoworld_amd import _lib
import tools.bench_kernels as B
lib = _lib.load()
for rnd in range(2):
  for dbg, name in ((512, "128x256 3-stage"), (0, "256x256 2-stage")):
    lib.ew_set_gemm_debug(dbg); print("##", name)
    B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
    B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
    B.gemm_case("L2 ff_up_geglu", 28800, 10240, 1280, act=2)
    B.gemm_case("L3 ff_up_geglu", 7200, 10240, 1280, act=2)
