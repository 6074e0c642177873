// elementwise.hip -- layout conversion at the U-Net boundary and the fused denoise-step glue (gfx950).
// HBM-bound; one pass each.  Reference: evoworld/pipeline/pipeline_evoworld.py:691-695 (CFG duplicate,
// scale_model_input, channel concat), :709-711 (CFG combine), :714 (EulerDiscreteScheduler.step, v-prediction).
#include "common.h"

namespace {

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int N, int C, int HW, int ldc,
                                    int c_off, float scale) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW;
        const int p = (int)(i - n * HW);
        const float* xp = x + n * C * (long long)HW + p;
        f16* yp = y + i * ldc + c_off;
        for (int c = 0; c < C; ++c) yp[c] = (f16)(scale * xp[(long long)c * HW]);
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
                                 int cpad, int T, int HW) {
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
        f16x4 o;
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
        }
        *(f16x4*)(nxt + i * cpad) = o;
        *(f16x4*)(nxt + (i + total) * cpad) = o;
    }
}

inline int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096); }

}  // namespace

extern "C" ew_status ew_nchw_f32_to_nhwc_f16(const float* x, void* y, int N, int C, int H, int W, int ldc, int c_off,
                                             float scale, void* stream) {
    EW_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && c_off >= 0 && c_off + C <= ldc, "ew_nchw_f32_to_nhwc_f16: bad args");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                       (f16*)y, N, C, H * W, ldc, c_off, scale);
    return ew_check_launch("ew_nchw_f32_to_nhwc_f16");
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
                       (const f16*)eps, ld_eps, latents, guidance, sigma, sigma_next, (f16*)next_in, cpad, T, h * w);
    return ew_check_launch("ew_euler_cfg_step");
}
