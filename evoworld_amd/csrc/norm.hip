// norm.hip -- GroupNorm (stats + apply[+SiLU]) and LayerNorm for channels-last fp16 activations (gfx950).
// HBM-bound kernels: 16-byte vector loads/stores, fp32 statistics, wave64 reductions.
// Reference call sites: torch.nn.GroupNorm / LayerNorm inside the diffusers blocks instantiated by
// evoworld/trainer/unet_plucker.py:161-233, conv_norm_out :236/478.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid = (row chunks, n_slabs); block = PL * VPP threads where VPP = C_src/8
// vectors per row and PL row-lanes.  Every thread owns one fixed 8-channel vector -> per-channel
// register accumulators -> LDS per-channel sums -> per-group reduce -> one fp32 atomic pair per
// (block, group).
// ---------------------------------------------------------------------------------------------
__global__ void gn_stats_kernel(const f16* __restrict__ x, float* __restrict__ sums, int rows, int C_src, int c_off,
                                int C_tot, int groups, int VPP, int PL, int rows_per_block) {
    extern __shared__ float lds[];  // [2][C_src]
    const int tid = threadIdx.x;
    const int slab = blockIdx.y;
    for (int i = tid; i < 2 * C_src; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const int v = tid % VPP, pl = tid / VPP;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const f16* base = x + ((size_t)slab * rows) * C_src + v * 8;
    for (int r = r0 + pl; r < r1; r += PL) {
        const f16x8 val = *(const f16x8*)(base + (size_t)r * C_src);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)val[e];
            s[e] += f;
            q[e] += f * f;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(&lds[v * 8 + e], s[e]);
        atomicAdd(&lds[C_src + v * 8 + e], q[e]);
    }
    __syncthreads();
    const int gs = C_tot / groups;
    const int g_lo = c_off / gs, g_hi = (c_off + C_src - 1) / gs;
    for (int g = g_lo + tid; g <= g_hi; g += blockDim.x) {
        const int c_lo = max(g * gs, c_off) - c_off, c_hi = min((g + 1) * gs, c_off + C_src) - c_off;
        float ss = 0.f, qq = 0.f;
        for (int c = c_lo; c < c_hi; ++c) { ss += lds[c]; qq += lds[C_src + c]; }
        atomicAdd(&sums[((size_t)slab * groups + g) * 2 + 0], ss);
        atomicAdd(&sums[((size_t)slab * groups + g) * 2 + 1], qq);
    }
}

__global__ void gn_apply_kernel(const f16* __restrict__ x, const float* __restrict__ sums, const f16* __restrict__ gamma,
                                const f16* __restrict__ beta, f16* __restrict__ y, long long n_vec, int rows, int C_src,
                                int c_off, int C_tot, int groups, float eps, int silu) {
    const int VPP = C_src / 8;
    const int gs = C_tot / groups;
    const float inv_cnt = 1.0f / ((float)rows * (float)gs);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / VPP;
        const int c = (int)(i - row * VPP) * 8;
        const int slab = (int)(row / rows);
        const f16x8 val = *(const f16x8*)(x + row * C_src + c);
        const f16x8 gm = *(const f16x8*)(gamma + c_off + c);
        const f16x8 bt = *(const f16x8*)(beta + c_off + c);
        f16x8 o;
        int gprev = -1;
        float mean = 0.f, rstd = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c_off + c + e) / gs;
            if (g != gprev) {
                gprev = g;
                const float s = sums[((size_t)slab * groups + g) * 2 + 0], q = sums[((size_t)slab * groups + g) * 2 + 1];
                mean = s * inv_cnt;
                const float var = fmaxf(q * inv_cnt - mean * mean, 0.f);
                rstd = rsqrtf(var + eps);
            }
            float f = ((float)val[e] - mean) * rstd * (float)gm[e] + (float)bt[e];
            if (silu) f = ew_silu(f);
            o[e] = (f16)f;
        }
        *(f16x8*)(y + row * C_tot + c_off + c) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (<= 4 vectors of 8 per lane => C <= 2048),
// two-pass (mean, then centred variance) on registers.
// ---------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void ln_kernel(const f16* __restrict__ x, const f16* __restrict__ addvec, int rpg,
                                                 f16* __restrict__ x_out, const f16* __restrict__ gamma,
                                                 const f16* __restrict__ beta, f16* __restrict__ y, int rows, int C,
                                                 float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int VPP = C / 8;
    float v[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int vi = lane + k * 64;
        if (vi < VPP) {
            const f16x8 t = *(const f16x8*)(x + (size_t)row * C + vi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = (float)t[e];
            if (addvec) {
                const f16x8 a = *(const f16x8*)(addvec + (size_t)(row / rpg) * C + vi * 8);
                f16x8 xo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // the sum is rounded to fp16 first: it IS the fp16 residual stream the block continues from
                    xo[e] = (f16)(v[k][e] + (float)a[e]);
                    v[k][e] = (float)xo[e];
                }
                if (x_out) *(f16x8*)(x_out + (size_t)row * C + vi * 8) = xo;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[k][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int vi = lane + k * 64;
        if (vi < VPP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int vi = lane + k * 64;
        if (vi < VPP) {
            const f16x8 g = *(const f16x8*)(gamma + vi * 8);
            const f16x8 b = *(const f16x8*)(beta + vi * 8);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)((v[k][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            *(f16x8*)(y + (size_t)row * C + vi * 8) = o;
        }
    }
}

}  // namespace

extern "C" ew_status ew_groupnorm_stats_f16(const void* x, float* sums, int n_slabs, int rows, int C_src, int c_off,
                                            int C_tot, int groups, void* stream) {
    EW_REQUIRE(x && sums, "ew_groupnorm_stats_f16: null pointer");
    EW_REQUIRE(n_slabs > 0 && rows > 0 && C_src > 0 && C_src % 8 == 0 && C_src <= 8192, "ew_groupnorm_stats_f16: bad shape");
    EW_REQUIRE(groups > 0 && C_tot % groups == 0 && c_off >= 0 && c_off + C_src <= C_tot, "ew_groupnorm_stats_f16: bad groups");
    const int VPP = C_src / 8;
    EW_REQUIRE(VPP <= 1024, "ew_groupnorm_stats_f16: C_src too large");
    const int PL = VPP >= 256 ? 1 : 256 / VPP;
    const int rpb = 64 * PL;
    dim3 grid(ew_cdiv(rows, rpb), n_slabs);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(VPP * PL), 2 * C_src * sizeof(float), (hipStream_t)stream,
                       (const f16*)x, sums, rows, C_src, c_off, C_tot, groups, VPP, PL, rpb);
    return ew_check_launch("ew_groupnorm_stats_f16");
}

extern "C" ew_status ew_groupnorm_apply_f16(const void* x, const float* sums, const void* gamma, const void* beta, void* y,
                                            int n_slabs, int rows, int C_src, int c_off, int C_tot, int groups, float eps,
                                            int silu, void* stream) {
    EW_REQUIRE(x && sums && gamma && beta && y, "ew_groupnorm_apply_f16: null pointer");
    EW_REQUIRE(n_slabs > 0 && rows > 0 && C_src > 0 && C_src % 8 == 0 && c_off % 8 == 0 && C_tot % 8 == 0,
               "ew_groupnorm_apply_f16: bad shape");
    EW_REQUIRE(groups > 0 && C_tot % groups == 0 && c_off >= 0 && c_off + C_src <= C_tot, "ew_groupnorm_apply_f16: bad groups");
    const long long n_vec = (long long)n_slabs * rows * (C_src / 8);
    const int blocks = (int)((n_vec + 255) / 256 < 8192 ? (n_vec + 255) / 256 : 8192);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)x, sums,
                       (const f16*)gamma, (const f16*)beta, (f16*)y, n_vec, rows, C_src, c_off, C_tot, groups, eps, silu);
    return ew_check_launch("ew_groupnorm_apply_f16");
}

extern "C" ew_status ew_layernorm_f16(const void* x, const void* addvec, int rows_per_group, void* x_out, const void* gamma,
                                      const void* beta, void* y, int rows, int C, float eps, void* stream) {
    EW_REQUIRE(x && gamma && beta && y, "ew_layernorm_f16: null pointer");
    EW_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048, "ew_layernorm_f16: need C %% 8 == 0 and C <= 2048 (C=%d)", C);
    EW_REQUIRE(!addvec || rows_per_group >= 1, "ew_layernorm_f16: rows_per_group must be >= 1");
    const int nv = (C / 8 + 63) / 64;
    dim3 grid(ew_cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int rpg = rows_per_group >= 1 ? rows_per_group : 1;
#define LN_LAUNCH(NV)                                                                                              \
    hipLaunchKernelGGL(ln_kernel<NV>, grid, block, 0, s, (const f16*)x, (const f16*)addvec, rpg, (f16*)x_out,      \
                       (const f16*)gamma, (const f16*)beta, (f16*)y, rows, C, eps)
    if (nv == 1) LN_LAUNCH(1);
    else if (nv == 2) LN_LAUNCH(2);
    else if (nv == 3) LN_LAUNCH(3);
    else LN_LAUNCH(4);
#undef LN_LAUNCH
    return ew_check_launch("ew_layernorm_f16");
}
