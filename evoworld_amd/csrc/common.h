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
// exact (erf) GELU, as torch.nn.functional.gelu default (diffusers GEGLU)
__device__ __forceinline__ float ew_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

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
