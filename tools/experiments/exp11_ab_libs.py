"""In-process A/B of two builds of the library: evoworld_amd/libevoworld_hip.so (working tree) against
evoworld_amd/libevoworld_hip_base.so (a build of an earlier commit, made with `git archive <rev> | make`).  Box speed differs
between gpurun calls, so only an alternating in-process comparison is trustworthy.
Usage: python tools/experiments/exp11_ab_libs.py [filter]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1] + sys.argv[1:]
from evoworld_amd import _lib, ops  # noqa: E402
import tools.bench_kernels as B  # noqa: E402

new = _lib.load()
base = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libevoworld_hip_base.so"))
base.ew_gemm_f16.argtypes = new.ew_gemm_f16.argtypes
base.ew_gemm_f16.restype = new.ew_gemm_f16.restype
new_fn, base_fn = new.ew_gemm_f16, base.ew_gemm_f16


def cases():
    B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2)
    B.gemm_case("L1 ff_up_geglu", 115200, 5120, 640, act=2)
    B.gemm_case("L2 ff_up_geglu", 28800, 10240, 1280, act=2)
    B.gemm_case("L0 ff_down_res", 460800, 320, 1280, res=True)
    B.gemm_case("L1 ff_down_res", 115200, 640, 2560, res=True)
    B.gemm_case("L2 ff_down_res", 28800, 1280, 5120, res=True)
    B.gemm_case("L0 proj_res", 460800, 320, 320, res=True)
    B.gemm_case("L1 proj_res", 115200, 640, 640, res=True)
    B.gemm_case("L1 qkv", 115200, 1920, 640)
    B.gemm_case("L0 qkv", 460800, 960, 320)
    B.conv_case("L0 320", 50, 320, 320, 72, 128)
    B.conv_case("L0 cat640", 50, 320, 320, 72, 128, c2=320)
    B.conv_case("L1 640", 50, 640, 640, 36, 64)
    B.conv_case("L1 up", 50, 640, 640, 36, 64, up=1)
    B.conv_case("L2 1280", 50, 1280, 1280, 18, 32)
    B.conv_case("L2 cat2560", 50, 1280, 1280, 18, 32, c2=1280)
    B.convt_case("L0", 2, 25, 9216, 320)
    B.convt_case("L2", 2, 25, 576, 1280)


for rnd in range(2):
    for name, fn in (("base", base_fn), ("new", new_fn)):
        new.ew_gemm_f16 = fn
        print("##", name, flush=True)
        cases()
new.ew_gemm_f16 = new_fn
