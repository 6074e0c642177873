import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0]]
from evoworld_amd import _lib
import tools.bench_kernels as B
lib = _lib.load()
for rnd in range(2):
  for dbg, name in ((0, "chunk-major"), (8, "tap-major")):
    lib.ew_set_gemm_debug(dbg); print("##", name)
    B.conv_case("L0 320", 50, 320, 320, 72, 128)
    B.conv_case("L0 cat960", 50, 640, 320, 72, 128, c2=320)
    B.conv_case("L1 640", 50, 640, 640, 36, 64)
    B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
    B.convt_case("L0", 2, 25, 9216, 320)
    B.convt_case("L2", 2, 25, 576, 1280)
