import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0]]
import tools.bench_kernels as B
B.attn_case("L0", 50, 9216, 5)
