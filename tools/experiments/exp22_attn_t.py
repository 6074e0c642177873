import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
import tools.bench_kernels as B
B.attn_t_case("L0", 2, 25, 9216, 5)
B.attn_t_case("L1", 2, 25, 2304, 10)
B.attn_t_case("L2", 2, 25, 576, 20)
B.attn_t_case("L3", 2, 25, 144, 20)
