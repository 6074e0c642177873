import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
import tools.bench_kernels as B
B.timeit.__defaults__ = (1, 0)
B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
B.gemm_case("L2 ff_up_geglu", 28800, 10240, 1280, act=2)
B.gemm_case("L0 qkv", 460800, 960, 320)
B.gemm_case("L2 ff_down_res", 28800, 1280, 5120, res=True)
B.conv_case("L0 320", 50, 320, 320, 72, 128)
B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
