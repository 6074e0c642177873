// elementwise.hip -- layout conversion at the U-Net boundary and the fused denoise-step glue (gfx950).
// HBM-bound; one pass each.  Reference: evoworld/pipeline/pipeline_evoworld.py:691-695 (CFG duplicate,
// scale_model_input, channel concat), :709-711 (CFG combine), :714 (EulerDiscreteScheduler.step, v-prediction).
#include "common.h"

namespace {

constexpr float SPLIT_DUP_SCALE = 0.0009765625f;   // 2^-10 (evoworld_amd/unet.py SPLIT_DUP_LOG2): the duplicate block meets W_lo * 2^10, a normal fp16 number

// lo_off > 0 (ABI 9, split operand): besides hi = fp16(v) at channel c_off + c the row also receives lo = fp16(v - hi) at lo_off + c_off + c and a
// copy of hi scaled by 2^-10 at dup_off + c_off + c -- the three K blocks [x_hi | x_lo | x_hi 2^-10] that conv_in multiplies with [W_hi | W_hi | W_lo 2^10]
// inside the 64-channel K tile it pads its 18 input channels to anyway (evoworld_amd/unet.py: conv_in with exact operands).
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int N, int C, int HW, int ldc,
                                    int c_off, float scale, int lo_off, int dup_off) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW;
        const int p = (int)(i - n * HW);
        const float* xp = x + n * C * (long long)HW + p;
        f16* yp = y + i * ldc + c_off;
        for (int c = 0; c < C; ++c) {
            const float v = scale * xp[(long long)c * HW];
            const f16 hi = (f16)v;
            yp[c] = hi;
            if (lo_off > 0) {
                yp[lo_off + c] = (f16)(v - (float)hi);
                yp[dup_off + c] = (f16)((float)hi * SPLIT_DUP_SCALE);
            }
        }
    }
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): out[r] = [cos(v f_j) | sin(v f_j)], f_j = exp(-ln(10000) j / half),
// v = vals[r % n_vals] (one timestep broadcast over the batch rows, or one added-time id per row).  fp32 math in the operation order of the torch
// expression it replaces (evoworld_amd/unet.py _sinusoid), rounded to fp16 once.
__global__ void sinusoid_kernel(const float* __restrict__ vals, int n_vals, int n_rows, int dim, f16* __restrict__ out) {
    const int half = dim / 2;
    const int total = n_rows * half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / half, j = i - r * half;
        const float f = expf((-9.210340371976184f * (float)j) / (float)half);
        const float a = vals[r % n_vals] * f;
        out[(size_t)r * dim + j] = (f16)cosf(a);
        out[(size_t)r * dim + half + j] = (f16)sinf(a);
    }
}

__global__ void nhwc_to_nchw_kernel(const f16* __restrict__ x, float* __restrict__ y, int N, int C, int HW, int ldc) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW;
        const int p = (int)(i - n * HW);
        const f16* xp = x + i * ldc;
        float* yp = y + n * C * (long long)HW + p;
        for (int c = 0; c < C; ++c) yp[(long long)c * HW] = (float)xp[c];
    }
}

// one thread per (frame t, pixel): 4 latent channels
__global__ void euler_cfg_kernel(const f16* __restrict__ eps, int ld_eps, float* __restrict__ lat,
                                 const float* __restrict__ guidance, float sigma, float sigma_next, f16* __restrict__ nxt,
                                 int cpad, int T, int HW, int lo_off, int dup_off) {
    const long long total = (long long)T * HW;
    const float s2p1 = sigma * sigma + 1.0f;
    const float c_out = -sigma / sqrtf(s2p1);
    const float dt = sigma_next - sigma;
    const float in_scale = 1.0f / sqrtf(sigma_next * sigma_next + 1.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / HW);
        const int p = (int)(i - (long long)t * HW);
        const f16x4 eu = *(const f16x4*)(eps + i * ld_eps);
        const f16x4 ec = *(const f16x4*)(eps + (i + total) * ld_eps);
        const float g = guidance[t];
        f16x4 o, ol, od;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float u = (float)eu[c], cn = (float)ec[c];
            const float e = u + g * (cn - u);                       // pipeline_evoworld.py:711
            float* lp = lat + ((long long)t * 4 + c) * HW + p;
            const float x = *lp;
            const float x0 = e * c_out + x / s2p1;                  // v-prediction (EulerDiscreteScheduler.step)
            const float d = (x - x0) / sigma;
            const float xn = x + d * dt;
            *lp = xn;
            o[c] = (f16)(xn * in_scale);                            // scale_model_input for the next step
            ol[c] = (f16)(xn * in_scale - (float)o[c]);             // split operand (lo_off > 0): what the fp16 rounding dropped
            od[c] = (f16)((float)o[c] * SPLIT_DUP_SCALE);
        }
        *(f16x4*)(nxt + i * cpad) = o;
        *(f16x4*)(nxt + (i + total) * cpad) = o;
        if (lo_off > 0) {
            *(f16x4*)(nxt + i * cpad + lo_off) = ol;
            *(f16x4*)(nxt + (i + total) * cpad + lo_off) = ol;
            *(f16x4*)(nxt + i * cpad + dup_off) = od;
            *(f16x4*)(nxt + (i + total) * cpad + dup_off) = od;
        }
    }
}

inline int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096); }

// Row softmax of a score matrix carried split (hi fp16 + lo8, common.h): p[r][c] = exp(s[r][c] - max_r) / sum_r, fp32 math, fp16 out.
// One wave per row, the row streams through registers in chunks of 8 columns per lane (two passes over the row: max+sum
// with the online rescale, then normalise; the second pass re-reads the row from L2).  Used by the VAE's single-head
// 512-wide attention (S = 9216 keys per frame): scores come from ew_gemm_f16 with out_lo, so nothing is rounded to fp16
// before the exponential (diffusers Attention upcast_softmax).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16* __restrict__ hi, const int8_t* __restrict__ lo,
                                                           f16* __restrict__ out, long long rows, int cols, long long ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f16* h = hi + row * ld;
    const int8_t* l = lo ? lo + row * ld : nullptr;
    float m = -INFINITY, sum = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        const f16x8 a = *(const f16x8*)(h + c);
        float v[8];
        if (l) {
            ew_split_dec8(a, *(const u32x2*)(l + c), v);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
        }
        float cm = v[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) cm = fmaxf(cm, v[e]);
        const float mn = fmaxf(m, cm);
        float cs = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) cs += __expf(v[e] - mn);
        sum = sum * __expf(m - mn) + cs;
        m = mn;
    }
    const float mw = wave_max(m);
    sum = wave_sum(sum * __expf(m - mw));
    const float inv = 1.0f / sum;
    f16* o = out + row * ld;
    for (int c = lane * 8; c < cols; c += 512) {
        const f16x8 a = *(const f16x8*)(h + c);
        float v[8];
        if (l) {
            ew_split_dec8(a, *(const u32x2*)(l + c), v);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
        }
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (f16)(__expf(v[e] - mw) * inv);
        *(f16x8*)(o + c) = r;
    }
}

// TemporalDecoder.time_conv_out: Conv3d(C, C, (3,1,1), padding (1,0,0)) on fp32 frames [B, T, C, HW] (C <= 4):
// y[b,t,o,p] = bias[o] + sum_{kt,c} w[o][c][kt] * x[b, t+kt-1, c, p], zero outside [0,T).  4 pixels per thread.
__global__ __launch_bounds__(256) void time_conv3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B, int T,
                                                         int C, int HW) {
    const int hq = HW / 4;
    const long long total = (long long)B * T * hq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p4 = (int)(i % hq);
        const long long bt = i / hq;
        const int t = (int)(bt % T);
        f32x4 in[3][4];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int tt = t + kt - 1;
            const bool ok = tt >= 0 && tt < T;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                in[kt][c] = (ok && c < C) ? *(const f32x4*)(x + ((bt + kt - 1) * C + c) * HW + (long long)p4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int o = 0; o < C; ++o) {
            f32x4 acc = {bias[o], bias[o], bias[o], bias[o]};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
                    if (c < C) acc += w[(o * C + c) * 3 + kt] * in[kt][c];
            *(f32x4*)(y + (bt * C + o) * HW + (long long)p4 * 4) = acc;
        }
    }
}

}  // namespace

extern "C" ew_status ew_nchw_f32_to_nhwc_f16(const float* x, void* y, int N, int C, int H, int W, int ldc, int c_off,
                                             float scale, void* stream) {
    EW_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && c_off >= 0 && c_off + C <= ldc, "ew_nchw_f32_to_nhwc_f16: bad args");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                       (f16*)y, N, C, H * W, ldc, c_off, scale, 0, 0);
    return ew_check_launch("ew_nchw_f32_to_nhwc_f16");
}

extern "C" ew_status ew_nchw_f32_to_nhwc_split_f16(const float* x, void* y, int N, int C, int H, int W, int ldc, int c_off, int lo_off,
                                                   int dup_off, float scale, void* stream) {
    EW_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && c_off >= 0, "ew_nchw_f32_to_nhwc_split_f16: bad args");
    EW_REQUIRE(lo_off >= c_off + C && dup_off >= lo_off + c_off + C && dup_off + c_off + C <= ldc,
               "ew_nchw_f32_to_nhwc_split_f16: the three channel blocks [c_off, +C), [lo_off + c_off, +C), [dup_off + c_off, +C) must be disjoint and inside ldc=%d", ldc);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                       (f16*)y, N, C, H * W, ldc, c_off, scale, lo_off, dup_off);
    return ew_check_launch("ew_nchw_f32_to_nhwc_split_f16");
}

extern "C" ew_status ew_sinusoid_embed_f16(const float* vals, int n_vals, int n_rows, int dim, void* out, void* stream) {
    EW_REQUIRE(vals && out && n_vals > 0 && n_rows > 0 && dim > 0 && dim % 2 == 0, "ew_sinusoid_embed_f16: bad args");
    const int total = n_rows * (dim / 2);
    hipLaunchKernelGGL(sinusoid_kernel, dim3(ew_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, vals, n_vals, n_rows, dim, (f16*)out);
    return ew_check_launch("ew_sinusoid_embed_f16");
}

extern "C" ew_status ew_nhwc_f16_to_nchw_f32(const void* x, float* y, int N, int C, int H, int W, int ldc, void* stream) {
    EW_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && C <= ldc, "ew_nhwc_f16_to_nchw_f32: bad args");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)x, y, N, C, H * W, ldc);
    return ew_check_launch("ew_nhwc_f16_to_nchw_f32");
}

extern "C" ew_status ew_euler_cfg_step(const void* eps, int ld_eps, float* latents, const float* guidance, float sigma,
                                       float sigma_next, void* next_in, int cpad, int T, int h, int w, void* stream) {
    EW_REQUIRE(eps && latents && guidance && next_in, "ew_euler_cfg_step: null pointer");
    EW_REQUIRE(T > 0 && h > 0 && w > 0 && ld_eps % 4 == 0 && cpad % 4 == 0 && ld_eps >= 4 && cpad >= 4, "ew_euler_cfg_step: bad shape");
    EW_REQUIRE(sigma > 0.f, "ew_euler_cfg_step: sigma must be > 0");
    hipLaunchKernelGGL(euler_cfg_kernel, dim3(grid_for((long long)T * h * w)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)eps, ld_eps, latents, guidance, sigma, sigma_next, (f16*)next_in, cpad, T, h * w, 0, 0);
    return ew_check_launch("ew_euler_cfg_step");
}

extern "C" ew_status ew_euler_cfg_step_split(const void* eps, int ld_eps, float* latents, const float* guidance, float sigma,
                                             float sigma_next, void* next_in, int cpad, int lo_off, int dup_off, int T, int h, int w,
                                             void* stream) {
    EW_REQUIRE(eps && latents && guidance && next_in, "ew_euler_cfg_step_split: null pointer");
    EW_REQUIRE(T > 0 && h > 0 && w > 0 && ld_eps % 4 == 0 && cpad % 4 == 0 && ld_eps >= 4 && cpad >= 4, "ew_euler_cfg_step_split: bad shape");
    EW_REQUIRE(lo_off >= 4 && lo_off % 4 == 0 && dup_off >= lo_off + 4 && dup_off % 4 == 0 && dup_off + 4 <= cpad,
               "ew_euler_cfg_step_split: lo_off / dup_off must be multiples of 4 with [0,4), [lo_off,+4), [dup_off,+4) disjoint inside cpad=%d", cpad);
    EW_REQUIRE(sigma > 0.f, "ew_euler_cfg_step_split: sigma must be > 0");
    hipLaunchKernelGGL(euler_cfg_kernel, dim3(grid_for((long long)T * h * w)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)eps, ld_eps, latents, guidance, sigma, sigma_next, (f16*)next_in, cpad, T, h * w, lo_off, dup_off);
    return ew_check_launch("ew_euler_cfg_step_split");
}

extern "C" ew_status ew_softmax_rows_f16(const void* hi, const void* lo, void* out, long long rows, int cols, long long ld,
                                         void* stream) {
    EW_REQUIRE(hi && out && rows > 0 && cols > 0, "ew_softmax_rows_f16: bad args");
    EW_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "ew_softmax_rows_f16: cols and ld must be multiples of 8");
    EW_REQUIRE((rows + 3) / 4 < 0x7fffffffLL, "ew_softmax_rows_f16: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const f16*)hi,
                       (const int8_t*)lo, (f16*)out, rows, cols, ld);
    return ew_check_launch("ew_softmax_rows_f16");
}

extern "C" ew_status ew_time_conv3_f32(const float* x, const float* w, const float* bias, float* y, int B, int T, int C, int HW,
                                       void* stream) {
    EW_REQUIRE(x && w && bias && y && B > 0 && T > 0 && C > 0 && C <= 4 && HW > 0 && HW % 4 == 0, "ew_time_conv3_f32: need C <= 4, HW %% 4 == 0");
    long long blocks = ((long long)B * T * (HW / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(time_conv3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, T, C, HW);
    return ew_check_launch("ew_time_conv3_f32");
}
