import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0]]
import tools.bench_kernels as B
B.attn_case("L0", 50, 9216, 5)
B.attn_case("L1", 50, 2304, 10)
B.conv_case("L0 320", 50, 320, 320, 72, 128)
B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
