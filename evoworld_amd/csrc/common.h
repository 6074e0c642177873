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
ew_status ew_check_launch(const char* what);

#define EW_REQUIRE(cond, ...)                    \
    do {                                         \
        if (!(cond)) {                           \
            ew_set_error(__VA_ARGS__);           \
            return EW_ERR_INVALID_ARG;           \
        }                                        \
    } while (0)

static inline int ew_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

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
// lanes-elements per issue); only the two transcendentals stay scalar.  Same A&S 7.1.26 erf as ew_erf, with the sign folded
// away: gelu(g) = 0.5 * (g + |g| * erf(|g| / sqrt 2)).  ~15.5 issue slots per element instead of ~24: the GEGLU epilogue of
// the K = 320 feed-forward GEMMs is one third of their run time.
__device__ __forceinline__ f32x2 ew_vgelu2(f32x2 v, f32x2 g) {
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
