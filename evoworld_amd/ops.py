"""Thin tensor->pointer wrappers over the C ABI (include/evoworld_hip.h).  torch is used only for device
memory and the current HIP stream; every op below is one hand-written HIP kernel launch.  No fallback."""
import ctypes

import torch

from . import _lib
from ._lib import GemmArgs

A_DENSE, A_CONV3X3, A_CONVT3 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_GELU = 0, 1, 2, 3

_zero_pages = {}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _aligned(t, n=16):
    """A contiguous tensor whose data pointer is n-byte aligned: `t` itself when it already is (the usual case: whole
    allocations), otherwise a fresh copy -- a frame-filtered slice such as conf[frames] or xyz[1:] of a tensor with
    H*W % 4 != 0 starts at an odd offset, and the 16-byte vector kernels need an aligned base."""
    t = t.contiguous()
    return t if t.data_ptr() % n == 0 else t.clone(memory_format=torch.contiguous_format)


def _req(t, dtype, name):
    if t.device.type != "cuda":
        raise _lib.EvoWorldHipError(f"{name} must live on the GPU (got {t.device}); evoworld_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def zero_page(device):
    key = str(device)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(4096, dtype=torch.float16, device=device)
    return _zero_pages[key]


class Res:
    """A residual-stream tensor carried split: hi = fp16(x) plus an int8 companion `lo` (one byte per element:
    bits(x) ~= bits(float(hi)) + 32 * lo on the fp32 bit patterns, ~19 mantissa bits in 3 bytes; include/evoworld_hip.h,
    ew_gemm_args), or lo = None when the stream is kept in plain fp16.  The reference runs the stream in fp32
    (unified_loop_consistency.py:188); consumers that need an fp16 MFMA operand read `hi` alone, the residual epilogues and
    the norms read both."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo=None):
        self.hi, self.lo = hi, lo

    @classmethod
    def empty(cls, rows, C, device, split):
        hi = torch.empty(rows, C, dtype=torch.float16, device=device)
        return cls(hi, torch.empty(rows, C, dtype=torch.int8, device=device) if split else None)

    @classmethod
    def from_float(cls, x):
        """fp32 tensor -> split form (host-side twin of the kernels' encoder: tests, debugging taps)."""
        x = x.float().contiguous()
        hi = x.half()
        d = (x.view(torch.int32) - hi.float().view(torch.int32) + 16) >> 5
        return cls(hi, d.clamp_(-128, 127).to(torch.int8))

    def float(self):
        if self.lo is None:
            return self.hi.float()
        return (self.hi.float().contiguous().view(torch.int32) + (self.lo.to(torch.int32) << 5)).view(torch.float32)


def _hl(x):
    """tensor | Res | None -> (hi, lo)"""
    if x is None:
        return None, None
    if isinstance(x, Res):
        return x.hi, x.lo
    return x, None


def gemm(a, w, out, *, M, N, c1, lda, a2=None, c2=0, lda2=0, bias=None, rowbias=None, rows_per_group=1, ld_rowbias=None,
         r1=None, ld_r1=0, r2=None, ld_r2=0, ld_out=None, mode=A_DENSE, conv=None, tconv=None, act=ACT_NONE,
         c_acc=1.0, c_r1=1.0, c_r2=1.0, conv_shift=0):
    """out = c_acc*act(A@W^T + bias + rowbias) + c_r1*r1 + c_r2*r2  (see ew_gemm_f16).
    conv = (n_img, h_in, w_in, h_out, w_out, stride, upsample); tconv = (B, T, P); conv_shift=1: padding (0,1) taps.
    r1 / r2 / out may be `Res` (split-fp16 residual stream): the lo halves ride along (ew_gemm_args.r1_lo ...)."""
    lib = _lib.load()
    g = GemmArgs()
    r1h, r1l = _hl(r1)
    r2h, r2l = _hl(r2)
    outh, outl = _hl(out)
    g.a, g.a2, g.w, g.bias, g.rowbias = _ptr(a), _ptr(a2), _ptr(w), _ptr(bias), _ptr(rowbias)
    g.r1, g.r2, g.out, g.zero_page = _ptr(r1h), _ptr(r2h), _ptr(outh), _ptr(zero_page(a.device))
    g.r1_lo, g.r2_lo, g.out_lo = _ptr(r1l), _ptr(r2l), _ptr(outl)
    g.M, g.N, g.c1, g.c2, g.lda, g.lda2 = M, N, c1, c2, lda, lda2
    n_out = N // 2 if act == ACT_GEGLU else N
    g.ld_out = ld_out if ld_out is not None else n_out
    g.ld_r1, g.ld_r2, g.mode = ld_r1, ld_r2, mode
    g.ld_rowbias = ld_rowbias if ld_rowbias is not None else N
    if conv is not None:
        g.n_img, g.h_in, g.w_in, g.h_out, g.w_out, g.stride, g.upsample = conv
    if tconv is not None:
        g.tB, g.tT, g.tP = tconv
    g.rows_per_group, g.act = rows_per_group, act
    g.c_acc, g.c_r1, g.c_r2 = c_acc, c_r1, c_r2
    g.conv_shift = conv_shift
    _lib.check(lib.ew_gemm_f16(ctypes.byref(g), _stream()), "ew_gemm_f16")
    return out


_sk_ready = set()


def streamk_init():
    """Best effort: allocate generation 3's stream-K workspace for (current device, current stream) once, outside any kernel
    launch path (ew_gemm_streamk_init).  Nothing depends on it succeeding -- a launch without a workspace runs the whole-tile
    schedule -- so a full pool (64 (device, stream) pairs per process) or a failed allocation only warns.  Skipped when the
    tail split cannot be used (generation != 3) and while the current stream is being captured into a graph
    (allocation + memset are illegal there; call it on the capture stream BEFORE the capture to get the tail inside the graph)."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    if key in _sk_ready:
        return
    if _lib.load().ew_get_gemm_generation() != 3:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    _sk_ready.add(key)
    if _lib.load().ew_gemm_streamk_init(_stream()) != 0:
        import warnings
        msg = _lib.load().ew_last_error()
        warnings.warn("ew_gemm_streamk_init: " + (msg.decode() if msg else "failed") + " (continuing without the stream-K tail on this stream)")


def streamk_check():
    """Raise if a stream-K finisher ever gave up waiting for its partner's partial (the tile it wrote is wrong).  Synchronises:
    call it where results are handed to the caller (end of a denoise loop / U-Net call / bench), not per launch."""
    st = _lib.load().ew_gemm_streamk_status()
    if st != 0:
        msg = _lib.load().ew_last_error()          # names the launch (kernel variant, M / N / K, stream) when the time-out was recorded
        raise _lib.EvoWorldHipError((msg.decode() if msg else "stream-K hand-over timed out in ew_gemm_f16 (generation 3): results of that launch are invalid")
                                    if st > 0 else "ew_gemm_streamk_status: HIP error while reading the status word")


def linear(x, w, bias=None, out=None, split_out=False, **kw):
    """x [M,K] fp16 (row stride = K), w [N,K] fp16 -> [M,N] (a `Res` with a lo half when split_out)."""
    _req(x, torch.float16, "x"); _req(w, torch.float16, "w")
    M, K = x.shape
    N = w.shape[0]
    act = kw.get("act", ACT_NONE)
    if out is None:
        n_out = N // 2 if act == ACT_GEGLU else N
        out = Res.empty(M, n_out, x.device, True) if split_out else torch.empty(M, n_out, dtype=torch.float16, device=x.device)
    return gemm(x, w, out, M=M, N=N, c1=K, lda=K, bias=bias, **kw)


class WorkspacePool:
    """fp32 scratch for the GroupNorm statistics of one forward (partials, pivots, stats: all written before they are read, so
    nothing is zeroed).  `reset()` rewinds; `take(n)` hands out the next n floats."""

    def __init__(self, device, floats=8 << 20):
        self.buf = torch.empty(floats, dtype=torch.float32, device=device)
        self.cur = 0

    def reset(self):
        self.cur = 0

    def take(self, n):
        n = (n + 63) // 64 * 64
        if self.cur + n > self.buf.numel():
            return torch.empty(n, dtype=torch.float32, device=self.buf.device)      # overflow: a fresh buffer
        v = self.buf[self.cur: self.cur + n]
        self.cur += n
        return v


SumsPool = WorkspacePool   # former name


def groupnorm(xs, gamma, beta, n_slabs, rows, eps, silu, groups=32, out=None, pool=None, stats_hi_only=False, split_out=False):
    """GroupNorm(+SiLU) over the channel concat of `xs` (list of [n_slabs*rows, C_i] fp16 tensors or `Res`) ->
    [n_slabs*rows, sum C_i] fp16.  Deterministic shifted statistics: stats per source, one finalize, apply per source.
    split_out: the result is the split operand [n_slabs*rows, 2 * sum C_i] = [y_hi | y_lo], y_lo = fp16(y - y_hi) (ew_groupnorm_apply_split_f16)."""
    lib = _lib.load()
    srcs = [_hl(x) for x in xs]
    C_tot = sum(h.shape[-1] for h, _ in srcs)
    dev = srcs[0][0].device
    nws = lib.ew_groupnorm_workspace_floats(n_slabs, rows, C_tot, groups)
    ws = pool.take(nws) if pool is not None else torch.empty(nws, dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty(n_slabs * rows, 2 * C_tot if split_out else C_tot, dtype=torch.float16, device=dev)
    st = _stream()
    off = 0
    for h, l in srcs:
        # stats_hi_only (off): statistics from the hi half alone would save the lo read of this pass (2.3 ms per forward), but
        # the rounding remainders add ulp^2/12 to the variance -- 0.5 % when a channel's std is ~4 fp16 ulps of its mean
        # (mean/std = 300), i.e. exactly the cancellation-prone inputs the shifted statistics exist for
        _lib.check(lib.ew_groupnorm_stats_f16(_ptr(h), None if stats_hi_only else _ptr(l), _ptr(ws), n_slabs, rows, h.shape[-1],
                                              off, C_tot, groups, st), "ew_groupnorm_stats_f16")
        off += h.shape[-1]
    _lib.check(lib.ew_groupnorm_finalize(_ptr(ws), n_slabs, rows, C_tot, groups, st), "ew_groupnorm_finalize")
    off = 0
    for h, l in srcs:
        if split_out:
            _lib.check(lib.ew_groupnorm_apply_split_f16(_ptr(h), _ptr(l), _ptr(ws), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(out[:, C_tot:]),
                                                        2 * C_tot, n_slabs, rows, h.shape[-1], off, C_tot, groups, eps, 1 if silu else 0, st),
                       "ew_groupnorm_apply_split_f16")
        else:
            _lib.check(lib.ew_groupnorm_apply_f16(_ptr(h), _ptr(l), _ptr(ws), _ptr(gamma), _ptr(beta), _ptr(out), n_slabs, rows,
                                                  h.shape[-1], off, C_tot, groups, eps, 1 if silu else 0, st),
                       "ew_groupnorm_apply_f16")
        off += h.shape[-1]
    return out


def layernorm(x, gamma, beta, eps=1e-5, addvec=None, rows_per_group=1, x_out=None, out=None):
    """x, x_out: fp16 tensors or `Res` (split-fp16 residual stream)."""
    lib = _lib.load()
    xh, xl = _hl(x)
    oh, ol = _hl(x_out)
    rows, C = xh.shape
    if out is None:
        out = torch.empty_like(xh)
    _lib.check(lib.ew_layernorm_f16(_ptr(xh), _ptr(xl), _ptr(addvec), rows_per_group, _ptr(oh), _ptr(ol), _ptr(gamma),
                                    _ptr(beta), _ptr(out), rows, C, eps, _stream()), "ew_layernorm_f16")
    return out


def attn_spatial(q, k, vt, o, n_seq, S, heads, ld_qk, ld_vt, ld_o, scale=0.125):
    lib = _lib.load()
    _lib.check(lib.ew_attn_spatial_f16(_ptr(q), _ptr(k), _ptr(vt), _ptr(o), n_seq, S, heads, ld_qk, ld_vt, ld_o, scale,
                                       _stream()), "ew_attn_spatial_f16")
    return o


QK_LOG2_PRESCALE = (0.125 * 1.4426950408889634) ** 0.5    # sqrt(head_dim^-0.5 * log2 e), head_dim 64: c_acc of the q|k projection


def attn_spatial_log2(q, k, vt, o, n_seq, S, heads, ld_qk, ld_vt, ld_o):
    """q, k pre-scaled by QK_LOG2_PRESCALE each (projection epilogue): the kernel's MFMA subtracts the running max itself."""
    lib = _lib.load()
    _lib.check(lib.ew_attn_spatial_log2_f16(_ptr(q), _ptr(k), _ptr(vt), _ptr(o), n_seq, S, heads, ld_qk, ld_vt, ld_o, _stream()),
               "ew_attn_spatial_log2_f16")
    return o


def attn_temporal(q, k, v, o, B, T, S, heads, ld, ld_o, scale=0.125):
    lib = _lib.load()
    _lib.check(lib.ew_attn_temporal_f16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, T, S, heads, ld, ld_o, scale, _stream()),
               "ew_attn_temporal_f16")
    return o


def softmax_rows(scores, out=None):
    """scores: fp16 [R, C] tensor or `Res` (hi + lo) -> fp16 softmax over the last dim (fp32 math)."""
    lib = _lib.load()
    h, l = _hl(scores)
    R, C = h.shape
    if out is None:
        out = torch.empty_like(h)
    _lib.check(lib.ew_softmax_rows_f16(_ptr(h), _ptr(l), _ptr(out), R, C, C, _stream()), "ew_softmax_rows_f16")
    return out


def time_conv3(x, w, bias):
    """x fp32 [B,T,C,H,W], w fp32 [C,C,3], bias fp32 [C] -> fp32 [B,T,C,H,W] (Conv3d (3,1,1), zero padding in T)."""
    lib = _lib.load()
    _req(x, torch.float32, "x"); _req(w, torch.float32, "w"); _req(bias, torch.float32, "bias")
    B, T, C, H, W = x.shape
    y = torch.empty_like(x)
    _lib.check(lib.ew_time_conv3_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, T, C, H * W, _stream()), "ew_time_conv3_f32")
    return y


def sinusoid_embed(vals, n_rows, dim):
    """vals fp32 device [n] -> fp16 [n_rows, dim] = [cos | sin] sinusoidal embedding of vals[row % n] (ew_sinusoid_embed_f16)."""
    lib = _lib.load()
    _req(vals, torch.float32, "vals")
    out = torch.empty(n_rows, dim, dtype=torch.float16, device=vals.device)
    _lib.check(lib.ew_sinusoid_embed_f16(_ptr(vals), vals.numel(), n_rows, dim, _ptr(out), _stream()), "ew_sinusoid_embed_f16")
    return out


def nchw_f32_to_nhwc_f16(x, y, ldc, c_off=0, scale=1.0, split=None):
    """split = (lo_off, dup_off): the row also receives fp16(v - hi) at lo_off + c_off + c and hi again at dup_off + c_off + c
    (ew_nchw_f32_to_nhwc_split_f16: the [x_hi | x_lo | x_hi] A operand of a conv_in packed [W_hi | W_hi | W_lo])."""
    lib = _lib.load()
    _req(x, torch.float32, "x"); _req(y, torch.float16, "y")
    N, C, H, W = x.shape
    if split:
        _lib.check(lib.ew_nchw_f32_to_nhwc_split_f16(_ptr(x), _ptr(y), N, C, H, W, ldc, c_off, split[0], split[1], scale, _stream()),
                   "ew_nchw_f32_to_nhwc_split_f16")
        return y
    _lib.check(lib.ew_nchw_f32_to_nhwc_f16(_ptr(x), _ptr(y), N, C, H, W, ldc, c_off, scale, _stream()),
               "ew_nchw_f32_to_nhwc_f16")
    return y


def nhwc_f16_to_nchw_f32(x, N, C, H, W, ldc):
    lib = _lib.load()
    _req(x, torch.float16, "x")
    y = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    _lib.check(lib.ew_nhwc_f16_to_nchw_f32(_ptr(x), _ptr(y), N, C, H, W, ldc, _stream()), "ew_nhwc_f16_to_nchw_f32")
    return y


def euler_cfg_step(eps, ld_eps, latents, guidance, sigma, sigma_next, next_in, cpad, T, h, w, split=None):
    lib = _lib.load()
    _req(latents, torch.float32, "latents"); _req(guidance, torch.float32, "guidance")
    if split:
        _lib.check(lib.ew_euler_cfg_step_split(_ptr(eps), ld_eps, _ptr(latents), _ptr(guidance), float(sigma), float(sigma_next),
                                               _ptr(next_in), cpad, split[0], split[1], T, h, w, _stream()), "ew_euler_cfg_step_split")
        return
    _lib.check(lib.ew_euler_cfg_step(_ptr(eps), ld_eps, _ptr(latents), _ptr(guidance), float(sigma), float(sigma_next),
                                     _ptr(next_in), cpad, T, h, w, _stream()), "ew_euler_cfg_step")


def plucker_embed(rays, c2w):
    lib = _lib.load()
    _req(rays, torch.float32, "rays"); _req(c2w, torch.float32, "c2w")
    H, W, _ = rays.shape
    N = c2w.shape[0]
    out = torch.empty(N, 6, H, W, dtype=torch.float32, device=rays.device)
    _lib.check(lib.ew_plucker_embed(_ptr(rays), _ptr(c2w), _ptr(out), N, H, W, _stream()), "ew_plucker_embed")
    return out


def cube2equi_gather(faces, lut, H, W):
    """faces uint8 [V,6,res,res,3|4] (order right,left,bottom,top,front,back), lut int16 [H,W,3] -> uint8 [V,H,W,3]."""
    lib = _lib.load()
    _req(faces, torch.uint8, "faces"); _req(lut, torch.int16, "lut")
    V, res, ch = faces.shape[0], faces.shape[2], faces.shape[-1]
    pano = torch.empty(V, H, W, 3, dtype=torch.uint8, device=faces.device)
    _lib.check(lib.ew_cube2equi_gather(_ptr(faces), ch, _ptr(lut), _ptr(pano), V, H, W, res, _stream()), "ew_cube2equi_gather")
    return pano


def select_kth(x, k):
    """x fp32 [n] on the device -> fp32 [2] device tensor (x_(k), x_(k+1)) (0-based, ascending): radix select, no sort."""
    lib = _lib.load()
    x = _aligned(x)
    _req(x, torch.float32, "x")
    ws = torch.empty(lib.ew_select_workspace_bytes() // 4 + 4, dtype=torch.int32, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.check(lib.ew_select_kth_f32(_ptr(x), x.numel(), int(k), _ptr(ws), _ptr(out), _stream()), "ew_select_kth_f32")
    return out


def filter_compact(conf, thr, xyz, img, img_nchw_hw=0):
    """conf fp32 [n], xyz fp32 [n,3], img fp32 [n,3] (or [S,3,hw] planes with img_nchw_hw = hw) ->
    (xyz_kept [m,3] fp32, rgbx [m,4] uint8 whose first 3 bytes are (img*255) truncated), order preserved."""
    lib = _lib.load()
    conf, xyz, img = _aligned(conf), _aligned(xyz), _aligned(img)
    _req(conf, torch.float32, "conf"); _req(xyz, torch.float32, "xyz"); _req(img, torch.float32, "img")
    n = conf.numel()
    dev = conf.device
    out_xyz = torch.empty(n, 3, dtype=torch.float32, device=dev)
    out_rgbx = torch.empty(n, 4, dtype=torch.uint8, device=dev)
    ws = torch.empty(lib.ew_filter_compact_workspace_bytes(n) // 4 + 1, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.ew_filter_compact(_ptr(conf), n, float(thr), _ptr(xyz), _ptr(img), 1 if img_nchw_hw else 0,
                                     int(img_nchw_hw), _ptr(out_xyz), _ptr(out_rgbx), _ptr(ws), _ptr(total), _stream()),
               "ew_filter_compact")
    m = int(total.item())
    return out_xyz[:m], out_rgbx[:m]


def depth_unproject(depth, extr, intr):
    lib = _lib.load()
    _req(depth, torch.float32, "depth"); _req(extr, torch.float32, "extr"); _req(intr, torch.float32, "intr")
    S, H, W = depth.shape
    xyz = torch.empty(S, H, W, 3, dtype=torch.float32, device=depth.device)
    _lib.check(lib.ew_depth_unproject(_ptr(depth), _ptr(extr), _ptr(intr), _ptr(xyz), S, H, W, _stream()),
               "ew_depth_unproject")
    return xyz


def splat_cubemap(xyz, rgb, w2c, res, fx, fy, cx, cy, z_near, face_channels=3):
    """xyz [N,3] f32, rgb u8 [N,3] (packed) or [N,4] (RGBX words, possibly a [:, :3] view of one), w2c [V,6,3,4] f32 ->
    faces u8 [V,6,res,res,face_channels], zbuf u64-as-int64 [V,6,res,res]."""
    lib = _lib.load()
    xyz = _aligned(xyz)
    _req(xyz, torch.float32, "xyz"); _req(w2c, torch.float32, "w2c")
    if rgb.dtype != torch.uint8 or rgb.device.type != "cuda":
        raise TypeError("rgb must be a uint8 device tensor")
    if rgb.ndim == 2 and rgb.stride(0) == 4 and rgb.stride(1) == 1:
        stride = 4                                                     # RGBX words (ew_filter_compact output)
    else:
        rgb, stride = rgb.contiguous(), 3
    V = w2c.shape[0]
    zbuf = torch.empty((V, 6, res, res), dtype=torch.int64, device=xyz.device)     # initialised by ew_splat_cubemap itself (0xFFFF... = no fragment)
    _lib.check(lib.ew_splat_cubemap(_ptr(xyz), xyz.shape[0], _ptr(w2c), _ptr(zbuf), V, res, fx, fy, cx, cy, z_near,
                                    _stream()), "ew_splat_cubemap")
    faces = torch.empty(V, 6, res, res, face_channels, dtype=torch.uint8, device=xyz.device)
    _lib.check(lib.ew_splat_resolve(_ptr(zbuf), _ptr(rgb), stride, _ptr(faces), face_channels, V, res, _stream()),
               "ew_splat_resolve")
    return faces, zbuf


def equi2pers(equi, rot, Hp, Wp, fov_x):
    lib = _lib.load()
    _req(equi, torch.uint8, "equi"); _req(rot, torch.float32, "rot")
    F_, He, We, _ = equi.shape
    out = torch.empty(F_, Hp, Wp, 3, dtype=torch.uint8, device=equi.device)
    _lib.check(lib.ew_equi2pers(_ptr(equi), _ptr(rot), _ptr(out), F_, He, We, Hp, Wp, float(fov_x), _stream()), "ew_equi2pers")
    return out


def resize_aa_u8(src, coeffs_h, coeffs_v, Ho, Wo):
    """src uint8 [V,Hi,Wi,3]; coeffs_* = (kk int32 [n_out,ksize], bounds int32 [n_out,2]) device tensors -> uint8 [V,Ho,Wo,3]."""
    lib = _lib.load()
    _req(src, torch.uint8, "src")
    V, Hi, Wi, _ = src.shape
    tmp = torch.empty(V, Hi, Wo, 3, dtype=torch.uint8, device=src.device)
    dst = torch.empty(V, Ho, Wo, 3, dtype=torch.uint8, device=src.device)
    (kh, bh), (kv, bv) = coeffs_h, coeffs_v
    _lib.check(lib.ew_resize_aa_u8(_ptr(src), _ptr(tmp), _ptr(dst), _ptr(kh), _ptr(bh), kh.shape[1], _ptr(kv), _ptr(bv),
                                   kv.shape[1], V, Hi, Wi, Ho, Wo, _stream()), "ew_resize_aa_u8")
    return dst


def u8_hwc_to_f32_chw(src):
    lib = _lib.load()
    _req(src, torch.uint8, "src")
    V, H, W, _ = src.shape
    dst = torch.empty(V, 3, H, W, dtype=torch.float32, device=src.device)
    _lib.check(lib.ew_u8_hwc_to_f32_chw(_ptr(src), _ptr(dst), V, H, W, _stream()), "ew_u8_hwc_to_f32_chw")
    return dst


def f32_chw_to_u8_hwc(src):
    """fp32 [V,3,H,W] in [-1,1] -> uint8 [V,H,W,3] (round-half-even of clamp(x/2+0.5,0,1)*255: the pipeline's PIL frames)."""
    lib = _lib.load()
    _req(src, torch.float32, "src")
    V, _, H, W = src.shape
    dst = torch.empty(V, H, W, 3, dtype=torch.uint8, device=src.device)
    _lib.check(lib.ew_f32_chw_to_u8_hwc(_ptr(src), _ptr(dst), V, H, W, _stream()), "ew_f32_chw_to_u8_hwc")
    return dst


def blur_axis(x, kern, axis):
    """x fp32 [..., H, W], kern fp32 [k] -> correlation along H (axis 0) or W (axis 1), reflect padding."""
    lib = _lib.load()
    _req(x, torch.float32, "x"); _req(kern, torch.float32, "kern")
    H, W = x.shape[-2:]
    out = torch.empty_like(x)
    _lib.check(lib.ew_blur_axis_f32(_ptr(x), _ptr(kern), kern.numel(), _ptr(out), x.numel() // (H * W), H, W, axis, _stream()),
               "ew_blur_axis_f32")
    return out


def bicubic_resize(x, Ho, Wo, scale=None, shift=None):
    """x fp32 [N,C,H,W] -> [N,C,Ho,Wo], bicubic align_corners=True; optional per-channel out = v*scale[c] + shift[c]."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    N, C, H, W = x.shape
    out = torch.empty(N, C, Ho, Wo, dtype=torch.float32, device=x.device)
    _lib.check(lib.ew_bicubic_resize_f32(_ptr(x), _ptr(out), N, C, H, W, Ho, Wo, _ptr(scale), _ptr(shift), _stream()),
               "ew_bicubic_resize_f32")
    return out


def vit_patchify(x, P, ldk):
    lib = _lib.load()
    _req(x, torch.float32, "x")
    N, _, S, _ = x.shape
    out = torch.empty(N * (S // P) ** 2, ldk, dtype=torch.float16, device=x.device)
    _lib.check(lib.ew_vit_patchify_f16(_ptr(x), _ptr(out), N, S, P, ldk, _stream()), "ew_vit_patchify_f16")
    return out


def attn_small(q, k, v, o, n_seq, S, heads, D, ld, ld_o, scale):
    lib = _lib.load()
    _lib.check(lib.ew_attn_small_f16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), n_seq, S, heads, D, ld, ld_o, float(scale), _stream()),
               "ew_attn_small_f16")
    return o


def ff_pack(w1, b1, w2):
    """Weights of one GEGLU feed-forward (w1 [2*H, C] value rows then gate rows, b1 [2*H], w2 [C, H]; fp32 or fp16, on the
    device) -> (w1p, b1p, w2p) fp16 in the LDS-image packs ew_ff_geglu320_f16 streams (layout: csrc/ff_fused.hip)."""
    dev = w1.device
    H2, C = w1.shape
    H = H2 // 2
    assert C == 320 and H == 1280 and tuple(w2.shape) == (C, H)
    nch = H // 32
    c = torch.arange(nch, device=dev)[:, None]
    r = torch.arange(64, device=dev)[None, :]
    j, i = r // 16, r % 16
    hh, vg = j // 2, j % 2
    src = vg * H + 32 * c + 8 * (i // 4) + 4 * hh + (i % 4)                                 # [nch, 64] proj rows
    sl = torch.arange(8, device=dev)
    w1g = w1.float()[src.reshape(-1)].reshape(nch, 64, C // 64, 8, 8)                       # [c, r, kt, slot, 8]
    perm1 = (sl[None, :] ^ (torch.arange(64, device=dev)[:, None] & 7))                     # packed slot sl <- k-slot sl ^ (r & 7)
    w1g = torch.gather(w1g, 3, perm1[None, :, None, :, None].expand(nch, 64, C // 64, 8, 8))
    w1p = w1g.permute(0, 2, 1, 3, 4).contiguous().to(torch.float16)                         # [c, kt, r, slot, 8]
    b1p = b1.float()[src.reshape(-1)].to(torch.float16).contiguous()
    r2 = torch.arange(C, device=dev)
    jj, i2 = r2 // 16, r2 % 16
    col = (jj // 2) * 32 + (i2 // 4) * 8 + (jj % 2) * 4 + (i2 % 4)                          # staged row -> output channel
    w2g = w2.float()[col].reshape(C, nch, 4, 8)                                              # [r, c, slot, 8]
    gq = torch.tensor([0, 2, 3, 1], device=dev)[(r2 >> 2) & 3]                              # bank-conflict-free slot XOR per row
    perm2 = (torch.arange(4, device=dev)[None, :] ^ gq[:, None])
    w2g = torch.gather(w2g, 2, perm2[:, None, :, None].expand(C, nch, 4, 8))
    w2p = w2g.permute(1, 0, 2, 3).contiguous().to(torch.float16)                             # [c, r, slot, 8]
    return w1p, b1p, w2p


def ff_geglu320(x, pack, b2, out, *, rowbias=None, rows_per_group=1, ld_rowbias=None, r1=None, r2=None, c_acc=1.0, c_r1=1.0, c_r2=1.0):
    """out = c_acc * (GEGLU(x W1^T + b1) W2^T + b2 + rowbias) + c_r1 * r1 + c_r2 * r2 for 320-channel tokens, one kernel
    (ew_ff_geglu320_f16): x fp16 [M, 320] (the LayerNorm output), pack = ff_pack(...); r1 / r2 / out tensors or `Res`."""
    lib = _lib.load()
    _req(x, torch.float16, "x")
    a = _lib.FfArgs()
    r1h, r1l = _hl(r1)
    r2h, r2l = _hl(r2)
    oh, ol = _hl(out)
    w1p, b1p, w2p = pack
    a.x, a.w1p, a.b1p, a.w2p, a.b2, a.rowbias = _ptr(x), _ptr(w1p), _ptr(b1p), _ptr(w2p), _ptr(b2), _ptr(rowbias)
    a.r1, a.r1_lo, a.r2, a.r2_lo, a.out, a.out_lo = _ptr(r1h), _ptr(r1l), _ptr(r2h), _ptr(r2l), _ptr(oh), _ptr(ol)
    a.zero_page = _ptr(zero_page(x.device))
    a.M, a.C, a.hidden = x.shape[0], x.shape[1], w2p.shape[0] * 32
    a.rows_per_group = rows_per_group
    a.ld_rowbias = ld_rowbias if ld_rowbias is not None else x.shape[1]
    a.c_acc, a.c_r1, a.c_r2 = c_acc, c_r1, c_r2
    _lib.check(lib.ew_ff_geglu320_f16(ctypes.byref(a), _stream()), "ew_ff_geglu320_f16")
    return out


def pack_conv_weight(w, cpad=None):
    """[O, I, *taps] (Conv2d 3x3 / Conv3d (3,1,1)) fp32 -> fp16 [O, K] in the K order ew_gemm_f16's conv modes read:
    [I/64 chunks][taps][64 channels].  `cpad` zero-pads the input channels first (conv_in: 18 -> 64)."""
    O, I = w.shape[0], w.shape[1]
    wt = w.reshape(O, I, -1).permute(0, 2, 1)                       # [O, taps, I]
    if cpad:
        wt = torch.nn.functional.pad(wt, (0, cpad - I))
        I = cpad
    taps = wt.shape[1]
    wt = wt.reshape(O, taps, I // 64, 64).permute(0, 2, 1, 3)       # [O, chunks, taps, 64]
    return wt.reshape(O, -1).to(torch.float16).contiguous()
