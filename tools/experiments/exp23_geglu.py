import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib
import tools.bench_kernels as B
new = _lib.load()
base = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libevoworld_hip_base.so"))
base.ew_gemm_f16.argtypes = new.ew_gemm_f16.argtypes; base.ew_gemm_f16.restype = new.ew_gemm_f16.restype
new_fn = new.ew_gemm_f16
for rnd in range(2):
    for name, fn in (("base", base.ew_gemm_f16), ("new", new_fn)):
        new.ew_gemm_f16 = fn
        print("##", name, flush=True)
        B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
        B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
        B.gemm_case("L2 ff_up_geglu", 28800, 10240, 1280, act=2)
        B.gemm_case("L3 ff_up_geglu", 7200, 10240, 1280, act=2)
new.ew_gemm_f16 = new_fn
