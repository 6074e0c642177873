import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
import tools.bench_kernels as B
B.timeit.__defaults__ = (2, 1)
B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
B.gemm_case("L2 ff_down_res", 28800, 1280, 5120, res=True)
