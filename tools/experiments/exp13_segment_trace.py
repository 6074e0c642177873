"""Where do the cycles of a K-tile go?  Runs GEMM cases on the instrumented build (make -C evoworld_amd/csrc trace) and prints,
per wave of workgroups 0 and 131, the mean shader cycles per stream position spent in:
  h0   = wait for fragments + ds_reads of the next half + 20 MFMAs (+ interleaved DMA pieces)   [issue time]
  vm   = s_waitcnt vmcnt (the DMA of K-tile v+1 landed)
  bar  = s_waitcnt lgkmcnt(0) + s_barrier
  h1   = ds_reads of K-tile v+1 + stage_begin + 20 MFMAs (+ DMA pieces)
  epi  = h1 of a tile-end position (includes the whole epilogue)
(s_memtime itself costs ~10 %: use for proportions, not absolutes.)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = sys.argv[:1]
from evoworld_amd import _lib  # noqa: E402
import tools.bench_kernels as B  # noqa: E402

new = _lib.load()
tr = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libevoworld_hip_trace.so"))
tr.ew_gemm_f16.argtypes = new.ew_gemm_f16.argtypes
tr.ew_gemm_f16.restype = new.ew_gemm_f16.restype
tr.ew_debug_trace_read.argtypes = [ctypes.c_void_p]
prod_fn = new.ew_gemm_f16
B.timeit.__defaults__ = (2, 1)


def show(title, run):
    new.ew_gemm_f16 = prod_fn
    print("-- product build:", end=" ")
    run()
    new.ew_gemm_f16 = tr.ew_gemm_f16
    print("-- traced build: ", end=" ")
    run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    tr.ew_debug_trace_read(buf)
    for blk in range(2):
        for w in range(8):
            o = buf[(blk * 8 + w) * 8:(blk * 8 + w + 1) * 8]
            V, nk, nte = o[6], o[7], o[5]
            if V == 0:
                continue
            n = V - 1
            print(f"   wg{'0' if blk == 0 else '131'} wave{w}: V={V} nk={nk} tiles={nte}  per position: h0 {o[0] / n:6.0f}  vm {o[1] / n:6.0f}  "
                  f"bar {o[2] / n:6.0f}  h1 {o[3] / max(1, n - nte):6.0f}  | tile-end h1+epilogue {o[4] / max(1, nte):7.0f}  "
                  f"| total/pos {(o[0] + o[1] + o[2] + o[3] + o[4]) / n:6.0f}")
    new.ew_gemm_f16 = prod_fn


show("conv L2", lambda: B.conv_case("L2 1280", 50, 1280, 1280, 18, 32))
show("dense L2 ff_down", lambda: B.gemm_case("L2 ff_down_res", 28800, 1280, 5120, res=True))
show("dense L0 qkv", lambda: B.gemm_case("L0 qkv", 460800, 960, 320))
show("geglu L0", lambda: B.gemm_case("L0 ff_up_geglu", 460800, 2560, 320, act=2))
