"""gen 3 main-loop variants (make -C evoworld_amd/csrc vars3): v1 = W prefetch distance 3, v2 = no sched_barrier pins, v3 = both."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib
import tools.bench_kernels as B
new = _lib.load()
prod = new.ew_gemm_f16
d = os.path.dirname(_lib.__file__)
libs = [("base", prod)]
for v in (1, 2, 3):
    L = ctypes.CDLL(os.path.join(d, f"libevoworld_hip_g3v{v}.so"))
    L.ew_gemm_f16.argtypes = prod.argtypes; L.ew_gemm_f16.restype = prod.restype
    libs.append((f"v{v}", L.ew_gemm_f16))
for rnd in range(2):
    for name, fn in libs:
        new.ew_gemm_f16 = fn
        print("##", name, flush=True)
        B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
        B.conv_case("L0 320", 50, 320, 320, 72, 128)
        B.gemm_case("L2 ff_down_res", 28800, 1280, 5120, res=True)
        B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
        B.gemm_case("L0 qkv", 460800, 960, 320)
new.ew_gemm_f16 = prod
