#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the distinct problem sizes of the config-2 U-Net forward.
Prints TFLOP/s (or GB/s) per problem; used to pick what to optimise.  Usage: python tools/bench_kernels.py [filter]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evoworld_amd import ops  # noqa: E402

DEV = "cuda"
FILTER = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*shape):
    return (torch.rand(*shape, device=DEV, dtype=torch.float16) * 2 - 1)


def report(name, ms, flops=None, bytes_=None):
    extra = []
    if flops:
        extra.append(f"{flops / ms / 1e9:8.1f} TF/s")
    if bytes_:
        extra.append(f"{bytes_ / ms / 1e6:8.1f} GB/s")
    print(f"{name:58s} {ms:9.3f} ms  " + "  ".join(extra), flush=True)


def gemm_case(name, M, N, K, act=0, res=False):
    if FILTER and FILTER not in "gemm " + name:
        return
    x, w, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
    out = torch.empty(M, N // 2 if act == 2 else N, dtype=torch.float16, device=DEV)
    r1 = rnd(M, N) if res else None
    fn = lambda: ops.gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=b, act=act, r1=r1, ld_r1=N if res else 0)
    report(f"gemm {name} M={M} N={N} K={K}", timeit(fn), 2.0 * M * N * K)


def conv_case(name, N, C, O, H, W, stride=1, up=0, c2=0):
    if FILTER and FILTER not in "conv3x3 " + name:
        return
    x = rnd(N * H * W, C)
    x2 = rnd(N * H * W, c2) if c2 else None
    Ho, Wo = (H // 2, W // 2) if stride == 2 else ((2 * H, 2 * W) if up else (H, W))
    w, b = rnd(O, 9 * (C + c2)) * 0.02, rnd(O)
    out = torch.empty(N * Ho * Wo, O, dtype=torch.float16, device=DEV)
    fn = lambda: ops.gemm(x, w, out, M=N * Ho * Wo, N=O, c1=C, lda=C, a2=x2, c2=c2, lda2=c2, bias=b, mode=ops.A_CONV3X3,
                          conv=(N, H, W, Ho, Wo, stride, up))
    report(f"conv3x3 {name} N={N} {C}+{c2}->{O} @{H}x{W} s{stride} up{up}", timeit(fn), 2.0 * N * Ho * Wo * O * 9 * (C + c2))


def convt_case(name, B, T, P, C):
    if FILTER and FILTER not in "convT3 " + name:
        return
    x, w, b = rnd(B * T * P, C), rnd(C, 3 * C) * 0.02, rnd(C)
    out = torch.empty(B * T * P, C, dtype=torch.float16, device=DEV)
    fn = lambda: ops.gemm(x, w, out, M=B * T * P, N=C, c1=C, lda=C, bias=b, mode=ops.A_CONVT3, tconv=(B, T, P))
    report(f"convT3 {name} B={B} T={T} P={P} C={C}", timeit(fn), 2.0 * B * T * P * C * 3 * C)


def attn_case(name, n_seq, S, heads):
    if FILTER and FILTER not in "attn_spatial " + name:
        return
    C = heads * 64
    rows = n_seq * S
    qk, vt = rnd(rows, 2 * C), rnd(C, rows)
    o = torch.empty(rows, C, dtype=torch.float16, device=DEV)
    fn = lambda: ops.attn_spatial(qk, qk[:, C:], vt, o, n_seq, S, heads, 2 * C, rows, C)
    report(f"attn_spatial {name} n={n_seq} S={S} h={heads}", timeit(fn, iters=3, warm=1), 4.0 * n_seq * heads * S * S * 64)


def attn_t_case(name, B, T, S, heads):
    if FILTER and FILTER not in "attn_temporal " + name:
        return
    C = heads * 64
    rows = B * T * S
    qkv = rnd(rows, 3 * C)
    o = torch.empty(rows, C, dtype=torch.float16, device=DEV)
    fn = lambda: ops.attn_temporal(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B, T, S, heads, 3 * C, C)
    report(f"attn_temporal {name} B={B} T={T} S={S} h={heads}", timeit(fn), 4.0 * B * S * heads * T * T * 64, rows * 4 * C * 2)


def norm_cases():
    for name, n, rows, C in (("gn2d L0", 50, 9216, 320), ("gn3d L0", 2, 230400, 320), ("gn2d L1", 50, 2304, 640), ("gn2d L0 960", 50, 9216, 960)):
        if FILTER and FILTER not in name:
            continue
        x, g, b = rnd(n * rows, C), rnd(C), rnd(C)
        out = torch.empty_like(x)
        fn = lambda: ops.groupnorm([x], g, b, n, rows, 1e-5, True, out=out)
        report(f"groupnorm {name}", timeit(fn), None, n * rows * C * 2 * 3)
    for name, rows, C in (("ln L0", 460800, 320), ("ln L1", 115200, 640), ("ln L2", 28800, 1280)):
        if FILTER and FILTER not in name:
            continue
        x, g, b = rnd(rows, C), rnd(C), rnd(C)
        out = torch.empty_like(x)
        fn = lambda: ops.layernorm(x, g, b, out=out)
        report(f"layernorm {name}", timeit(fn), None, rows * C * 2 * 2)


if __name__ == "__main__":
    t0 = time.time()
    for lvl, (tok, C) in enumerate(((460800, 320), (115200, 640), (28800, 1280), (7200, 1280))):
        gemm_case(f"L{lvl} ff_up_geglu", tok, 8 * C, C, act=2)
        gemm_case(f"L{lvl} ff_down_res", tok, C, 4 * C, res=True)
        gemm_case(f"L{lvl} proj_CxC_res", tok, C, C, res=True)
        gemm_case(f"L{lvl} qk", tok, 2 * C, C)
        gemm_case(f"L{lvl} qkv", tok, 3 * C, C)
        gemm_case(f"L{lvl} vT", C, tok, C)
    conv_case("L0 320", 50, 320, 320, 72, 128)
    conv_case("L0 cat640", 50, 320, 320, 72, 128, c2=320)
    conv_case("L0 cat960", 50, 640, 320, 72, 128, c2=320)
    conv_case("L0 out4", 50, 320, 4, 72, 128)
    conv_case("L0 in64", 50, 64, 320, 72, 128)
    conv_case("L0 down", 50, 320, 320, 72, 128, stride=2)
    conv_case("L1 640", 50, 640, 640, 36, 64)
    conv_case("L1 up", 50, 640, 640, 36, 64, up=1)
    conv_case("L2 1280", 50, 1280, 1280, 18, 32)
    conv_case("L2 cat2560", 50, 1280, 1280, 18, 32, c2=1280)
    conv_case("L3 1280", 50, 1280, 1280, 9, 16)
    convt_case("L0", 2, 25, 9216, 320)
    convt_case("L1", 2, 25, 2304, 640)
    convt_case("L2", 2, 25, 576, 1280)
    convt_case("L3", 2, 25, 144, 1280)
    attn_case("L0", 50, 9216, 5)
    attn_case("L1", 50, 2304, 10)
    attn_case("L2", 50, 576, 20)
    attn_case("L3", 50, 144, 20)
    attn_t_case("L0", 2, 25, 9216, 5)
    attn_t_case("L2", 2, 25, 576, 20)
    norm_cases()
    print(f"# total wall {time.time() - t0:.1f} s")
