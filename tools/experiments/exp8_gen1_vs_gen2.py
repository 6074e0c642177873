import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0]]
from evoworld_amd import _lib
import tools.bench_kernels as B
lib = _lib.load()
for rnd in range(2):
  for gen in (1, 2):
    lib.ew_set_gemm_generation(gen); print("## gen", gen)
    B.conv_case("L0 320", 50, 320, 320, 72, 128)
    B.conv_case("L0 cat960", 50, 640, 320, 72, 128, c2=320)
    B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
    B.convt_case("L0", 2, 25, 9216, 320)
    B.gemm_case("L0 qkv", 460800, 960, 320)
    B.gemm_case("L2 ff_down", 28800, 1280, 5120, res=True)
