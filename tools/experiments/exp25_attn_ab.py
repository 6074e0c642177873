import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib
import tools.bench_kernels as B
new = _lib.load()
base = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libevoworld_hip_base.so"))
base.ew_attn_spatial_f16.argtypes = new.ew_attn_spatial_f16.argtypes; base.ew_attn_spatial_f16.restype = new.ew_attn_spatial_f16.restype
new_fn = new.ew_attn_spatial_f16
for rnd in range(2):
    for name, fn in (("base", base.ew_attn_spatial_f16), ("new", new_fn)):
        new.ew_attn_spatial_f16 = fn
        print("##", name, flush=True)
        B.attn_case("L0", 50, 9216, 5)
        B.attn_case("L1", 50, 2304, 10)
        B.attn_case("L2", 50, 576, 20)
new.ew_attn_spatial_f16 = new_fn
