// common.h -- shared device/host helpers for libevoworld_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "evoworld_hip.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local error message (ew_last_error)
void ew_set_error(const char* fmt, ...);
int ew_cu_budget();                   // CUs the persistent kernels may assume (runtime.cpp; 256 unless ew_set_cu_budget changed it)
ew_status ew_check_launch(const char* what);

#define EW_REQUIRE(cond, ...)                    \
    do {                                         \
        if (!(cond)) {                           \
            ew_set_error(__VA_ARGS__);           \
            return EW_ERR_INVALID_ARG;           \
        }                                        \
    } while (0)

static inline int ew_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: cache "already set" per (kernel, device), thread-safe.
// `mask` is one static std::atomic<unsigned long long> per kernel instantiation (bit d = set on device d; devices >= 64 set it on
// every launch).
#include <atomic>
static inline ew_status ew_ensure_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 64; }
    if (dev < 64 && (mask.load(std::memory_order_acquire) >> dev) & 1ULL) return EW_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { ew_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return EW_ERR_HIP; }
    if (dev < 64) mask.fetch_or(1ULL << dev, std::memory_order_release);
    return EW_OK;
}

__device__ __forceinline__ float ew_silu(float x) { return x / (1.0f + __expf(-x)); }
// erf GELU (torch.nn.functional.gelu default, diffusers GEGLU).  erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7,
// i.e. below fp32 erff's own error after the fp16 output rounding): 2 transcendentals + ~10 VALU instead of libm's
// ~50-instruction erff -- the GEGLU epilogue was costing as much as the K=320 MFMA loop (profiles/r01 notes).
__device__ __forceinline__ float ew_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float y = fmaf(-p * t, e, 1.0f);
    return copysignf(y, x);
}
__device__ __forceinline__ float ew_gelu(float x) { return 0.5f * x * (1.0f + ew_erf(x * 0.70710678118654752440f)); }
// value * gelu(gate) for two (value, gate) pairs at once on the packed-fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32: two
// lane-elements per issue); only the transcendentals stay scalar.
// Round 3: gelu(g) = g * Phi(g) with Phi(g) ~= 1 / (1 + exp(-g * (c1 + c3 g^2 + c5 g^4))), the logistic form with an odd
// polynomial exponent, minimax-fitted to the erf GELU (torch.nn.functional.gelu default, diffusers GEGLU) on |g| <= 9 (g^2 is
// clamped at 64; beyond |g| = 8 the result is g or 0 to fp32 precision either way): max |gelu error| 2.6e-5 absolute (fp32
// evaluation included), i.e. 1/19 of the fp16 ulp at 1.0 the result is rounded to -- measured effect on the forward's rel-L2
// against the fp32 oracle: none (tests/test_gpu_unet.py).  13 issue slots per PAIR instead of the 31 of the A&S 7.1.26 erf form
// used before (ew_vgelu2_erf, kept for A/B: -DEW_GEGLU_ERF): the GEGLU epilogue of the K = 320 feed-forward GEMMs was a third of
// their run time.
__device__ __forceinline__ f32x2 ew_vgelu2_erf(f32x2 v, f32x2 g) {
    const f32x2 ag = {fabsf(g[0]), fabsf(g[1])};
    const f32x2 ax = ag * 0.70710678118654752440f;
    const f32x2 den = __builtin_elementwise_fma(ax, (f32x2){0.3275911f, 0.3275911f}, (f32x2){1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 p = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, (f32x2){1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, (f32x2){-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, (f32x2){0.254829592f, 0.254829592f});
    const f32x2 x2 = (ax * -1.4426950408889634f) * ax;
    const f32x2 e = {__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])};
    const f32x2 y = __builtin_elementwise_fma(-(p * t), e, (f32x2){1.0f, 1.0f});       // erf(|g| / sqrt 2)
    return (v * 0.5f) * __builtin_elementwise_fma(ag, y, g);
}
__device__ __forceinline__ f32x2 ew_vgelu2(f32x2 v, f32x2 g) {
    // k_i = -log2(e) * c_i, c = (1.59501577, 7.40112921e-2, -7.03033591e-4)
    f32x2 x2 = g * g;
    x2 = (f32x2){fminf(x2[0], 64.0f), fminf(x2[1], 64.0f)};
    f32x2 p = __builtin_elementwise_fma(x2, (f32x2){0.0010142630f, 0.0010142630f}, (f32x2){-0.10677572f, -0.10677572f});
    p = __builtin_elementwise_fma(p, x2, (f32x2){-2.3011212f, -2.3011212f});
    const f32x2 t = g * p;
    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + (f32x2){1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return (v * g) * r;
}

// ---- split residual stream: value = (hi fp16, lo8 int8) ------------------------------------------------------------------
// hi = fp16(x) (round to nearest even: the half a consumer can feed to an fp16 MFMA as it is); lo8 = the distance from hi to x
// in steps of 32 fp32 ulps, counted on the fp32 BIT PATTERN: bits(x) ~= bits((float)hi) + 32 * lo8.  |bits(x) - bits(hi)| <=
// 2^12 for any x that rounds to a normal hi (also across a binade boundary, where the integer order of IEEE patterns keeps
// encode and decode consistent), so lo8 in [-128, 127] carries 8 more mantissa bits (~19 in all) in ONE byte per element:
// 3 bytes per stream element instead of the 4 of a second fp16 plane, decode = cvt + bfe + shift-add (no more VALU work than
// the two cvt + add of an fp16 pair).  Tiny values (fp16-subnormal hi) lose the extra bits -- irrelevant for rel-L2.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ew_split_dec(f16 hi, int lo8) {
    return __int_as_float(__float_as_int((float)hi) + (lo8 << 5));
}
__device__ __forceinline__ int ew_split_enc(float x, f16 hi) {
    const int d = (__float_as_int(x) - __float_as_int((float)hi) + 16) >> 5;
    return min(max(d, -128), 127);
}
__device__ __forceinline__ int ew_sbyte(unsigned w, int i) { return (int)__builtin_amdgcn_sbfe((int)w, 8 * i, 8); }
__device__ __forceinline__ void ew_split_dec8(const f16x8 hi, const u32x2 lo, float (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = ew_split_dec(hi[e], ew_sbyte(lo[e >> 2], e & 3));
}
__device__ __forceinline__ unsigned ew_pack4(int a, int b, int c, int d) {
    return (unsigned)(a & 255) | ((unsigned)(b & 255) << 8) | ((unsigned)(c & 255) << 16) | ((unsigned)d << 24);
}
// x[8] fp32 -> hi (fp16 x 8) + lo8 (8 bytes)
__device__ __forceinline__ void ew_split_enc8(const float (&x)[8], f16x8& hi, u32x2& lo) {
    int s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = (f16)x[e]; s[e] = ew_split_enc(x[e], hi[e]); }
    lo[0] = ew_pack4(s[0], s[1], s[2], s[3]);
    lo[1] = ew_pack4(s[4], s[5], s[6], s[7]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
