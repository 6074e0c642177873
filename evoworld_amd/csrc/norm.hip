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
    // eight independent 16-byte loads in flight per thread (one dependent load per iteration ran at 3.3 TB/s, four at 3.8)
    int r = r0 + pl;
    for (; r + 7 * PL < r1; r += 8 * PL) {
        f16x8 val[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) val[u] = *(const f16x8*)(base + (size_t)(r + u * PL) * C_src);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)val[u][e];
                s[e] += f;
                q[e] += f * f;
            }
    }
    for (; r < r1; r += PL) {
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

// GroupNorm apply (+SiLU).  Same thread layout as the statistics kernel: grid = (row chunks, n_slabs), block = PL * VPP
// threads, every thread owns one fixed 8-channel vector of one slab -> mean / rstd / gamma / beta collapse ONCE per thread
// into scale[8], shift[8]; the row loop is load, 8 fma (+SiLU), store.  (The first version re-derived (row, column, slab,
// group) with 64-bit divisions for every vector and ran at 2.8 TB/s.)
__global__ void gn_apply_kernel(const f16* __restrict__ x, const float* __restrict__ sums, const f16* __restrict__ gamma,
                                const f16* __restrict__ beta, f16* __restrict__ y, int rows, int C_src, int c_off, int C_tot,
                                int groups, float eps, int silu, int VPP, int PL, int rows_per_block) {
    const int tid = threadIdx.x;
    const int slab = blockIdx.y;
    const int v = tid % VPP, pl = tid / VPP;
    const int c = v * 8;
    const int gs = C_tot / groups;
    const float inv_cnt = 1.0f / ((float)rows * (float)gs);
    const f16x8 gm = *(const f16x8*)(gamma + c_off + c);
    const f16x8 bt = *(const f16x8*)(beta + c_off + c);
    float scale[8], shift[8];
    int gprev = -1;
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c_off + c + e) / gs;
        if (g != gprev) {
            gprev = g;
            const float sm = sums[((size_t)slab * groups + g) * 2 + 0], q = sums[((size_t)slab * groups + g) * 2 + 1];
            mean = sm * inv_cnt;
            const float var = fmaxf(q * inv_cnt - mean * mean, 0.f);
            rstd = rsqrtf(var + eps);
        }
        // same evaluation order as before: ((x - mean) * rstd) * gamma + beta
        scale[e] = rstd;
        shift[e] = mean;
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    const f16* xin = x + ((size_t)slab * rows) * C_src + c;
    f16* yout = y + ((size_t)slab * rows) * C_tot + c_off + c;
    int r = r0 + pl;
    for (; r + 3 * PL < r1; r += 4 * PL) {
        f16x8 val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) val[u] = *(const f16x8*)(xin + (size_t)(r + u * PL) * C_src);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = ((float)val[u][e] - shift[e]) * scale[e] * (float)gm[e] + (float)bt[e];
                if (silu) f = ew_silu(f);
                o[e] = (f16)f;
            }
            *(f16x8*)(yout + (size_t)(r + u * PL) * C_tot) = o;
        }
    }
    for (; r < r1; r += PL) {
        const f16x8 val = *(const f16x8*)(xin + (size_t)r * C_src);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = ((float)val[e] - shift[e]) * scale[e] * (float)gm[e] + (float)bt[e];
            if (silu) f = ew_silu(f);
            o[e] = (f16)f;
        }
        *(f16x8*)(yout + (size_t)r * C_tot) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (<= 4 vectors of 8 per lane => C <= 2048),
// two-pass (mean, then centred variance) on registers.
// ---------------------------------------------------------------------------------------------
template <int NV, int RPW>
__global__ __launch_bounds__(256) void ln_kernel(const f16* __restrict__ x, const f16* __restrict__ addvec, int rpg,
                                                 f16* __restrict__ x_out, const f16* __restrict__ gamma,
                                                 const f16* __restrict__ beta, f16* __restrict__ y, int rows, int C,
                                                 float eps) {
    // a wave owns RPW consecutive rows and issues all their loads before reducing any of them: at C = 320 a row is only
    // 40 of the wave's 64 lanes wide, and one row at a time left a single 640-byte request in flight per wave (3.5 TB/s)
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int VPP = C / 8;
    float v[RPW][NV][8];
    f16x8 raw[RPW][NV], add[RPW][NV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, rows - 1);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int vi = lane + k * 64;
            if (vi < VPP) {
                raw[r][k] = *(const f16x8*)(x + (size_t)row * C + vi * 8);
                if (addvec) add[r][k] = *(const f16x8*)(addvec + (size_t)(row / rpg) * C + vi * 8);
            }
        }
    }
    f16x8 g[NV], bta[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int vi = lane + k * 64;
        if (vi < VPP) { g[k] = *(const f16x8*)(gamma + vi * 8); bta[k] = *(const f16x8*)(beta + vi * 8); }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        const bool live = row < rows;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int vi = lane + k * 64;
            if (vi < VPP) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[r][k][e] = (float)raw[r][k][e];
                if (addvec) {
                    f16x8 xo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // the sum is rounded to fp16 first: it IS the fp16 residual stream the block continues from
                        xo[e] = (f16)(v[r][k][e] + (float)add[r][k][e]);
                        v[r][k][e] = (float)xo[e];
                    }
                    if (x_out && live) *(f16x8*)(x_out + (size_t)row * C + vi * 8) = xo;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += v[r][k][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[r][k][e] = 0.f;
            }
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int vi = lane + k * 64;
            if (vi < VPP) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[r][k][e] - mean; sq += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int vi = lane + k * 64;
            if (vi < VPP && live) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)((v[r][k][e] - mean) * rstd * (float)g[k][e] + (float)bta[k][e]);
                *(f16x8*)(y + (size_t)row * C + vi * 8) = o;
            }
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
    int rpb = 128 * PL;                                        // rows per block: long streams (8 loads in flight x 16 iterations), but at least ~200 blocks
    while (rpb > 16 * PL && (long long)ew_cdiv(rows, rpb) * n_slabs < 200) rpb /= 2;
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
    const int VPP = C_src / 8;
    EW_REQUIRE(VPP <= 1024, "ew_groupnorm_apply_f16: C_src too large");
    const int PL = VPP >= 256 ? 1 : 256 / VPP;
    const int rpb = 32 * PL;                                   // 32 vectors in flight per thread-column
    dim3 grid(ew_cdiv(rows, rpb), n_slabs);
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(VPP * PL), 0, (hipStream_t)stream, (const f16*)x, sums,
                       (const f16*)gamma, (const f16*)beta, (f16*)y, rows, C_src, c_off, C_tot, groups, eps, silu, VPP, PL, rpb);
    return ew_check_launch("ew_groupnorm_apply_f16");
}

extern "C" ew_status ew_layernorm_f16(const void* x, const void* addvec, int rows_per_group, void* x_out, const void* gamma,
                                      const void* beta, void* y, int rows, int C, float eps, void* stream) {
    EW_REQUIRE(x && gamma && beta && y, "ew_layernorm_f16: null pointer");
    EW_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048, "ew_layernorm_f16: need C %% 8 == 0 and C <= 2048 (C=%d)", C);
    EW_REQUIRE(!addvec || rows_per_group >= 1, "ew_layernorm_f16: rows_per_group must be >= 1");
    const int nv = (C / 8 + 63) / 64;
    dim3 block(256);
    hipStream_t s = (hipStream_t)stream;
    const int rpg = rows_per_group >= 1 ? rows_per_group : 1;
#define LN_LAUNCH(NV, RPW)                                                                                         \
    hipLaunchKernelGGL((ln_kernel<NV, RPW>), dim3(ew_cdiv(rows, 4 * RPW)), block, 0, s, (const f16*)x,             \
                       (const f16*)addvec, rpg, (f16*)x_out, (const f16*)gamma, (const f16*)beta, (f16*)y, rows, C, eps)
    if (nv == 1) LN_LAUNCH(1, 4);
    else if (nv == 2) LN_LAUNCH(2, 4);
    else if (nv == 3) LN_LAUNCH(3, 2);
    else LN_LAUNCH(4, 2);
#undef LN_LAUNCH
    return ew_check_launch("ew_layernorm_f16");
}
