// clip.hip -- kernels of the CLIP image-encoder path: the reference's antialiased resize to 224x224
// (evoworld/pipeline/pipeline_evoworld.py:746-850: separable Gaussian blur with reflect padding, then bicubic interpolation
// with align_corners=True), the ViT patch embedding's im2col, and a small-sequence attention core for head_dim 80
// (ViT-H/14: 257 tokens x 16 heads).  All of it is tiny next to the denoise loop (one 224x224 image per clip): plain
// coalesced fp32 VALU kernels, no MFMA reshaping.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {       // torch 'reflect' padding (no edge repeat)
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// 1-D correlation along W (axis 1) or H (axis 0) with reflect padding: out[p] = sum_k kern[k] * in[reflect(p + k - pad_front)]
__global__ __launch_bounds__(256) void blur_axis_kernel(const float* __restrict__ x, const float* __restrict__ kern, int ksize,
                                                        float* __restrict__ out, long long planes, int H, int W, int axis) {
    const long long total = planes * H * W;
    const int pad_front = (ksize - 1) / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xw = (int)(i % W);
        const long long t = i / W;
        const int yh = (int)(t % H);
        const long long pl = t / H;
        const float* base = x + pl * H * W;
        float acc = 0.f;
        for (int k = 0; k < ksize; ++k) {
            const int yy = axis == 0 ? reflect_idx(yh + k - pad_front, H) : yh;
            const int xx = axis == 1 ? reflect_idx(xw + k - pad_front, W) : xw;
            acc += kern[k] * base[(long long)yy * W + xx];
        }
        out[i] = acc;
    }
}

// torch upsample_bicubic2d coefficients (A = -0.75)
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x1 = t + 1.0f, x2 = t, x3 = 1.0f - t, x4 = 2.0f - t;
    c[0] = ((A * x1 - 5.0f * A) * x1 + 8.0f * A) * x1 - 4.0f * A;
    c[1] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[2] = ((A + 2.0f) * x3 - (A + 3.0f)) * x3 * x3 + 1.0f;
    c[3] = ((A * x4 - 5.0f * A) * x4 + 8.0f * A) * x4 - 4.0f * A;
}

// F.interpolate(mode='bicubic', align_corners=True) + per-channel affine: out = v * scale[c] + shift[c]
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ x, float* __restrict__ out, long long planes,
                                                      int C, int H, int W, int Ho, int Wo, const float* __restrict__ scale,
                                                      const float* __restrict__ shift) {
    const long long total = planes * Ho * Wo;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const long long t = i / Wo;
        const int oy = (int)(t % Ho);
        const long long pl = t / Ho;
        const float ry = sy * (float)oy, rx = sx * (float)ox;
        const int iy = (int)floorf(ry), ix = (int)floorf(rx);
        float cy[4], cx[4];
        cubic_coeffs(ry - (float)iy, cy);
        cubic_coeffs(rx - (float)ix, cx);
        const float* base = x + pl * H * W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(ix - 1 + b, 0), W - 1);
                row += cx[b] * base[(long long)yy * W + xx];
            }
            acc += cy[a] * row;
        }
        const int c = (int)(pl % C);
        out[i] = scale ? acc * scale[c] + shift[c] : acc;
    }
}

// ViT patch embedding im2col: pixel_values fp32 [N,3,S,S] -> fp16 [N*(S/P)^2, ldk], K order (c, ky, kx) = the flattening of
// Conv2d(3, D, P, stride P).weight [D,3,P,P]; columns [3*P*P, ldk) are zero.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, f16* __restrict__ out, int N, int S, int P,
                                                       int ldk) {
    const int G = S / P, K = 3 * P * P;
    const long long total = (long long)N * G * G * ldk;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int kk = (int)(i % ldk);
        const long long row = i / ldk;
        float v = 0.f;
        if (kk < K) {
            const int c = kk / (P * P), r = kk - c * P * P, ky = r / P, kx = r - ky * P;
            const int n = (int)(row / (G * G)), g = (int)(row - (long long)n * G * G), gy = g / G, gx = g - gy * G;
            v = x[(((long long)n * 3 + c) * S + gy * P + ky) * S + gx * P + kx];
        }
        out[i] = (f16)v;
    }
}

// Small-sequence attention: one wave per (sequence, head, query).  q,k,v: fp16 token-major rows (row stride ld, head h at
// +h*D); D <= 128, D % 8 == 0... handled generically: lanes stride over the D dims for q.k and over keys for the softmax.
// scores live in LDS per wave (S <= 1024).  fp32 math throughout.
__global__ __launch_bounds__(256) void attn_small_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                         const f16* __restrict__ v, f16* __restrict__ o, int n_seq, int S,
                                                         int heads, int D, int ld, int ld_o, float scale) {
    extern __shared__ float lds[];                     // [4 waves][S scores + D q]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wid = (long long)blockIdx.x * 4 + wave;
    const long long total = (long long)n_seq * heads * S;
    if (wid >= total) return;
    const int qi = (int)(wid % S);
    const long long t = wid / S;
    const int h = (int)(t % heads);
    const long long seq = t / heads;
    float* sc = lds + (size_t)wave * (S + D);
    float* qs = sc + S;
    const f16* qp = q + (seq * S + qi) * ld + h * D;
    for (int d = lane; d < D; d += 64) qs[d] = (float)qp[d] * scale;
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
    for (int j = lane; j < S; j += 64) {
        const f16* kp = k + (seq * S + j) * ld + h * D;
        float acc = 0.f;
        for (int d = 0; d < D; d += 8) {
            const f16x8 kv = *(const f16x8*)(kp + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += qs[d + e] * (float)kv[e];
        }
        sc[j] = acc;
        m = fmaxf(m, acc);
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < S; j += 64) {
        const float p = __expf(sc[j] - m);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / sum;
    f16* op = o + (seq * S + qi) * ld_o + h * D;
    for (int d = lane; d < D; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += sc[j] * (float)v[(seq * S + j) * ld + h * D + d];
        op[d] = (f16)(acc * inv);
    }
}

}  // namespace

extern "C" ew_status ew_blur_axis_f32(const float* x, const float* kern, int ksize, float* out, long long planes, int H, int W,
                                      int axis, void* stream) {
    EW_REQUIRE(x && kern && out && planes > 0 && H > 0 && W > 0 && ksize > 0 && (axis == 0 || axis == 1), "ew_blur_axis_f32: bad args");
    EW_REQUIRE(ksize / 2 < (axis == 0 ? H : W), "ew_blur_axis_f32: kernel wider than reflect padding allows");
    long long b = (planes * H * W + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(blur_axis_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, kern, ksize, out, planes, H, W, axis);
    return ew_check_launch("ew_blur_axis_f32");
}

extern "C" ew_status ew_bicubic_resize_f32(const float* x, float* out, int N, int C, int H, int W, int Ho, int Wo,
                                           const float* scale, const float* shift, void* stream) {
    EW_REQUIRE(x && out && N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "ew_bicubic_resize_f32: bad args");
    EW_REQUIRE((scale == nullptr) == (shift == nullptr), "ew_bicubic_resize_f32: scale and shift go together");
    long long b = ((long long)N * C * Ho * Wo + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(bicubic_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, out, (long long)N * C, C, H, W, Ho, Wo,
                       scale, shift);
    return ew_check_launch("ew_bicubic_resize_f32");
}

extern "C" ew_status ew_vit_patchify_f16(const float* x, void* out, int N, int S, int P, int ldk, void* stream) {
    EW_REQUIRE(x && out && N > 0 && S > 0 && P > 0 && S % P == 0 && ldk >= 3 * P * P && ldk % 8 == 0, "ew_vit_patchify_f16: bad args");
    long long b = ((long long)N * (S / P) * (S / P) * ldk + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, (f16*)out, N, S, P, ldk);
    return ew_check_launch("ew_vit_patchify_f16");
}

extern "C" ew_status ew_attn_small_f16(const void* q, const void* k, const void* v, void* o, int n_seq, int S, int heads, int D,
                                       int ld, int ld_o, float scale, void* stream) {
    EW_REQUIRE(q && k && v && o && n_seq > 0 && S > 0 && heads > 0, "ew_attn_small_f16: bad args");
    EW_REQUIRE(D > 0 && D % 8 == 0 && D <= 256 && S <= 2048 && ld % 8 == 0, "ew_attn_small_f16: need D %% 8 == 0, D <= 256, S <= 2048");
    const long long total = (long long)n_seq * heads * S;
    const size_t lds = 4 * (size_t)(S + D) * sizeof(float);
    hipLaunchKernelGGL(attn_small_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), lds, (hipStream_t)stream, (const f16*)q,
                       (const f16*)k, (const f16*)v, (f16*)o, n_seq, S, heads, D, ld, ld_o, scale);
    return ew_check_launch("ew_attn_small_f16");
}
