// norm.hip -- GroupNorm (stats + apply[+SiLU]) and LayerNorm for channels-last fp16 activations (gfx950).
// HBM-bound kernels: 16-byte vector loads/stores, fp32 statistics, wave64 reductions.
// Reference call sites: torch.nn.GroupNorm / LayerNorm inside the diffusers blocks instantiated by
// evoworld/trainer/unet_plucker.py:161-233, conv_norm_out :236/478.
#include "common.h"
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics: deterministic and cancellation-safe.
//   stage 1 (gn_stats_kernel): grid = (row chunks, n_slabs); block = PL * VPP threads (VPP = C_src/8 vectors per row, PL
//     row-lanes).  Every thread owns one fixed 8-channel vector and accumulates SHIFTED sums  s = sum(x - K_c),
//     q = sum((x - K_c)^2)  with the pivot K_c = x[slab, row 0, c] (the same for every block of the slab, so partials add up;
//     |mean_c - K_c| ~ std_c, which keeps q - s^2/n free of the E[x^2] - mean^2 cancellation).  The PL row-lanes of a block
//     are combined through LDS in a FIXED order and the block writes one (s, q) pair per channel to
//     part[slab][chunk][C_tot][2] -- no atomics anywhere.
//   stage 2 (gn_finalize_kernel): one wave per (slab, group): lanes own channels, loop over the chunks in order, then a
//     fixed shuffle tree: mean_g, var_g (two-level: per-channel mean / M2, then across the group's channels).
// The chunking depends on (rows, n_slabs) only, so the two sources of a virtual channel concat share one partial buffer.
// x_lo (may be null): lo8 companion of a split residual stream (common.h: one byte per element, same element strides).
// ---------------------------------------------------------------------------------------------
inline int gn_chunks(int n_slabs, int rows) {
    constexpr int target = 640;
    int want = (target + n_slabs - 1) / n_slabs;               // ~2.5 blocks per CU over the whole launch
    const int cap = (rows + 63) / 64;                          // at least 64 rows per chunk
    if (want > cap) want = cap;
    if (want > 512) want = 512;
    return want < 1 ? 1 : want;
}

__global__ void gn_stats_kernel(const f16* __restrict__ x, const int8_t* __restrict__ x_lo, float* __restrict__ part,
                                float* __restrict__ pivot, int rows, int C_src, int c_off, int C_tot, int VPP, int PL,
                                int rows_per_block) {
    extern __shared__ float lds[];  // [PL][2][C_src]
    const int tid = threadIdx.x;
    const int slab = blockIdx.y;
    const int v = tid % VPP, pl = tid / VPP;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float s[8], q[8], piv[8];
    const size_t slab_off = ((size_t)slab * rows) * C_src + v * 8;
    const f16* base = x + slab_off;
    const int8_t* base_lo = x_lo ? x_lo + slab_off : nullptr;
    {
        const f16x8 p0 = *(const f16x8*)base;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; piv[e] = (float)p0[e]; }
        if (base_lo) ew_split_dec8(p0, *(const u32x2*)base_lo, piv);
    }
    // eight independent 16-byte loads in flight per thread (one dependent load per iteration ran at 3.3 TB/s, four at 3.8)
    int r = r0 + pl;
    if (base_lo) {
        constexpr int UL = 8;                       // rows in flight per thread on the hi + lo8 path (round 4 A/B, 4 vs 8: level-0 statistics pass 109 -> 102.5 us, profiles/r04_f_sweeps.txt)
        for (; r + (UL - 1) * PL < r1; r += UL * PL) {
            f16x8 val[UL];
            u32x2 vlo[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                val[u] = *(const f16x8*)(base + (size_t)(r + u * PL) * C_src);
                vlo[u] = *(const u32x2*)(base_lo + (size_t)(r + u * PL) * C_src);
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                float xv[8];
                ew_split_dec8(val[u], vlo[u], xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = xv[e] - piv[e];
                    s[e] += f;
                    q[e] += f * f;
                }
            }
        }
        for (; r < r1; r += PL) {
            const f16x8 val = *(const f16x8*)(base + (size_t)r * C_src);
            float xv[8];
            ew_split_dec8(val, *(const u32x2*)(base_lo + (size_t)r * C_src), xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = xv[e] - piv[e];
                s[e] += f;
                q[e] += f * f;
            }
        }
    } else {
        for (; r + 7 * PL < r1; r += 8 * PL) {
            f16x8 val[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) val[u] = *(const f16x8*)(base + (size_t)(r + u * PL) * C_src);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)val[u][e] - piv[e];
                    s[e] += f;
                    q[e] += f * f;
                }
        }
        for (; r < r1; r += PL) {
            const f16x8 val = *(const f16x8*)(base + (size_t)r * C_src);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)val[e] - piv[e];
                s[e] += f;
                q[e] += f * f;
            }
        }
    }
    float* mine = lds + (size_t)pl * 2 * C_src;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mine[v * 8 + e] = s[e];
        mine[C_src + v * 8 + e] = q[e];
    }
    __syncthreads();
    float* dst = part + (((size_t)slab * gridDim.x + blockIdx.x) * C_tot + c_off) * 2;
    for (int c = tid; c < C_src; c += blockDim.x) {
        float ss = 0.f, qq = 0.f;
        for (int k = 0; k < PL; ++k) { ss += lds[(size_t)k * 2 * C_src + c]; qq += lds[(size_t)k * 2 * C_src + C_src + c]; }
        *(f32x2*)(dst + 2 * c) = (f32x2){ss, qq};
    }
    if (blockIdx.x == 0 && pl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) pivot[(size_t)slab * C_tot + c_off + v * 8 + e] = piv[e];
    }
}

// one 256-thread block per (slab, group): stats[slab][g] = (mean, biased variance).  Thread (kc, c) sums the chunks
// kc, kc+KP, ... of channel c in order; the KP chunk-lanes of a channel are combined in order through LDS; the group's
// channels go through a fixed shuffle tree.  (One wave per group looping over all chunks took 29 us per call.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ pivot,
                                                          float* __restrict__ stats, int n_slabs, int rows, int C_tot,
                                                          int groups, int chunks) {
    __shared__ float red[2][256];
    __shared__ float chan[2][512];             // per-channel mean, M2 (gs <= 512)
    __shared__ float wsum[4];
    const int tid = threadIdx.x;
    const int slab = blockIdx.x / groups, g = blockIdx.x - slab * groups;
    const int gs = C_tot / groups;
    const float n = (float)rows;
    const int cw = gs < 256 ? gs : 256;        // channels handled per pass
    const int KP = 256 / cw;                   // chunk-lanes per channel
    for (int c0 = 0; c0 < gs; c0 += cw) {
        const int c = c0 + tid % cw, kc = tid / cw;
        float ss = 0.f, qq = 0.f;
        if (kc < KP && c < gs) {
            const float* p = part + (((size_t)slab * chunks) * C_tot + g * gs + c) * 2;
            for (int k = kc; k < chunks; k += KP) {
                const f32x2 v = *(const f32x2*)(p + (size_t)k * C_tot * 2);
                ss += v[0];
                qq += v[1];
            }
        }
        red[0][tid] = ss;
        red[1][tid] = qq;
        __syncthreads();
        if (tid < cw && c0 + tid < gs) {
            float s2 = 0.f, q2 = 0.f;
            for (int k = 0; k < KP; ++k) { s2 += red[0][k * cw + tid]; q2 += red[1][k * cw + tid]; }
            const float d = s2 / n;
            chan[0][c0 + tid] = pivot[(size_t)slab * C_tot + g * gs + c0 + tid] + d;     // channel mean
            chan[1][c0 + tid] = fmaxf(q2 - s2 * d, 0.f);                                  // channel M2
        }
        __syncthreads();
    }
    // across the group's channels: mean = avg(mean_c); M2 = sum(M2_c) + n * sum((mean_c - mean)^2)
    float ms = 0.f, m2 = 0.f;
    for (int c = tid; c < gs; c += 256) { ms += chan[0][c]; m2 += chan[1][c]; }
    ms = wave_sum(ms);
    m2 = wave_sum(m2);
    if ((tid & 63) == 0) wsum[tid >> 6] = ms;
    __syncthreads();
    const float mean = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)gs;
    __syncthreads();
    float dev = 0.f;
    for (int c = tid; c < gs; c += 256) { const float d = chan[0][c] - mean; dev += d * d; }
    dev = wave_sum(dev);
    if ((tid & 63) == 0) { wsum[tid >> 6] = m2 + n * dev; }
    __syncthreads();
    if (tid == 0) *(f32x2*)(stats + (size_t)blockIdx.x * 2) = (f32x2){mean, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (n * (float)gs)};
}

// GroupNorm apply (+SiLU).  Same thread layout as the statistics kernel: grid = (row chunks, n_slabs), block = PL * VPP
// threads, every thread owns one fixed 8-channel vector of one slab -> mean / rstd / gamma / beta collapse ONCE per thread
// into scale[8], shift[8]; the row loop is load, 8 fma (+SiLU), store.  (The first version re-derived (row, column, slab,
// group) with 64-bit divisions for every vector and ran at 2.8 TB/s.)
// YLO: the result is written as a SPLIT OPERAND (round 6): y = fp16(f) and y_lo = fp16(f - float(y)) with the same row stride ld_y -- the
// [x_hi | x_lo] A operand of a consumer whose weights are packed [W_hi | W_hi | W_lo] (conv_out, the level-0 proj_in).
template <bool LO, bool YLO>
__global__ void gn_apply_kernel(const f16* __restrict__ x, const int8_t* __restrict__ x_lo, const float* __restrict__ stats,
                                const f16* __restrict__ gamma, const f16* __restrict__ beta, f16* __restrict__ y,
                                f16* __restrict__ y_lo, int ld_y, int rows,
                                int C_src, int c_off, int C_tot, int groups, float eps, int silu, int VPP, int PL,
                                int rows_per_block) {
    const int tid = threadIdx.x;
    const int slab = blockIdx.y;
    const int v = tid % VPP, pl = tid / VPP;
    const int c = v * 8;
    const int gs = C_tot / groups;
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
            const f32x2 st = *(const f32x2*)(stats + ((size_t)slab * groups + g) * 2);
            mean = st[0];
            rstd = rsqrtf(st[1] + eps);
        }
        // evaluation order: ((x - mean) * rstd) * gamma + beta
        scale[e] = rstd;
        shift[e] = mean;
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    const f16* xin = x + ((size_t)slab * rows) * C_src + c;
    const int8_t* xlo = LO ? x_lo + ((size_t)slab * rows) * C_src + c : nullptr;
    f16* yout = y + ((size_t)slab * rows) * ld_y + c_off + c;
    f16* ylo = YLO ? y_lo + ((size_t)slab * rows) * ld_y + c_off + c : nullptr;
    int r = r0 + pl;
    for (; r + 3 * PL < r1; r += 4 * PL) {
        f16x8 val[4];
        u32x2 vlo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            val[u] = *(const f16x8*)(xin + (size_t)(r + u * PL) * C_src);
            if constexpr (LO) vlo[u] = *(const u32x2*)(xlo + (size_t)(r + u * PL) * C_src);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f16x8 o, ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xv;
                if constexpr (LO) xv = ew_split_dec(val[u][e], ew_sbyte(vlo[u][e >> 2], e & 3));
                else xv = (float)val[u][e];
                float f = (xv - shift[e]) * scale[e] * (float)gm[e] + (float)bt[e];
                if (silu) f = ew_silu(f);
                o[e] = (f16)f;
                if constexpr (YLO) ol[e] = (f16)(f - (float)o[e]);
            }
            *(f16x8*)(yout + (size_t)(r + u * PL) * ld_y) = o;
            if constexpr (YLO) *(f16x8*)(ylo + (size_t)(r + u * PL) * ld_y) = ol;
        }
    }
    for (; r < r1; r += PL) {
        const f16x8 val = *(const f16x8*)(xin + (size_t)r * C_src);
        u32x2 vlo;
        if constexpr (LO) vlo = *(const u32x2*)(xlo + (size_t)r * C_src);
        f16x8 o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float xv;
            if constexpr (LO) xv = ew_split_dec(val[e], ew_sbyte(vlo[e >> 2], e & 3));
            else xv = (float)val[e];
            float f = (xv - shift[e]) * scale[e] * (float)gm[e] + (float)bt[e];
            if (silu) f = ew_silu(f);
            o[e] = (f16)f;
            if constexpr (YLO) ol[e] = (f16)(f - (float)o[e]);
        }
        *(f16x8*)(yout + (size_t)r * ld_y) = o;
        if constexpr (YLO) *(f16x8*)(ylo + (size_t)r * ld_y) = ol;
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (<= 4 vectors of 8 per lane => C <= 2048),
// two-pass (mean, then centred variance) on registers.
// ---------------------------------------------------------------------------------------------
template <int NV, int RPW>
__global__ __launch_bounds__(256) void ln_kernel(const f16* __restrict__ x, const int8_t* __restrict__ x_lo,
                                                 const f16* __restrict__ addvec, int rpg, f16* __restrict__ x_out,
                                                 int8_t* __restrict__ x_out_lo, const f16* __restrict__ gamma,
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
    u32x2 rlo[RPW][NV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, rows - 1);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int vi = lane + k * 64;
            if (vi < VPP) {
                raw[r][k] = *(const f16x8*)(x + (size_t)row * C + vi * 8);
                if (x_lo) rlo[r][k] = *(const u32x2*)(x_lo + (size_t)row * C + vi * 8);
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
                if (x_lo) {
                    ew_split_dec8(raw[r][k], rlo[r][k], v[r][k]);                        // split stream: x = (hi, lo8)
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[r][k][e] = (float)raw[r][k][e];
                }
                if (addvec) {
                    f16x8 xo;
                    int s8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // what is normalised IS the residual stream the block continues from: fp16(x + add), or its
                        // (hi, lo8) split when the caller keeps the lo half (x_out_lo)
                        const float sum = v[r][k][e] + (float)add[r][k][e];
                        xo[e] = (f16)sum;
                        s8[e] = ew_split_enc(sum, xo[e]);
                        v[r][k][e] = !x_out ? sum : (x_out_lo ? ew_split_dec(xo[e], s8[e]) : (float)xo[e]);   // not materialised: exact
                    }
                    if (x_out && live) *(f16x8*)(x_out + (size_t)row * C + vi * 8) = xo;
                    if (x_out_lo && live)
                        *(u32x2*)(x_out_lo + (size_t)row * C + vi * 8) =
                            (u32x2){ew_pack4(s8[0], s8[1], s8[2], s8[3]), ew_pack4(s8[4], s8[5], s8[6], s8[7])};
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

extern "C" size_t ew_groupnorm_workspace_floats(int n_slabs, int rows, int C_tot, int groups) {
    if (n_slabs <= 0 || rows <= 0 || C_tot <= 0 || groups <= 0) return 0;
    // partials [n_slabs][chunks][C_tot][2] | pivots [n_slabs][C_tot] | stats [n_slabs][groups][2]
    return (size_t)n_slabs * gn_chunks(n_slabs, rows) * C_tot * 2 + (size_t)n_slabs * C_tot + (size_t)n_slabs * groups * 2;
}

namespace {
struct GnWs { float* part; float* pivot; float* stats; int chunks; };
inline GnWs gn_ws(float* ws, int n_slabs, int rows, int C_tot) {
    GnWs w;
    w.chunks = gn_chunks(n_slabs, rows);
    w.part = ws;
    w.pivot = ws + (size_t)n_slabs * w.chunks * C_tot * 2;
    w.stats = w.pivot + (size_t)n_slabs * C_tot;
    return w;
}
}  // namespace

extern "C" ew_status ew_groupnorm_stats_f16(const void* x, const void* x_lo, float* ws, int n_slabs, int rows, int C_src,
                                            int c_off, int C_tot, int groups, void* stream) {
    EW_REQUIRE(x && ws, "ew_groupnorm_stats_f16: null pointer");
    EW_REQUIRE(n_slabs > 0 && rows > 0 && C_src > 0 && C_src % 8 == 0 && C_src <= 8192, "ew_groupnorm_stats_f16: bad shape");
    EW_REQUIRE(groups > 0 && C_tot % groups == 0 && c_off >= 0 && c_off + C_src <= C_tot, "ew_groupnorm_stats_f16: bad groups");
    const int VPP = C_src / 8;
    EW_REQUIRE(VPP <= 1024, "ew_groupnorm_stats_f16: C_src too large");
    const int PL = VPP >= 256 ? 1 : 256 / VPP;
    const GnWs w = gn_ws(ws, n_slabs, rows, C_tot);
    const int rpb = ew_cdiv(rows, w.chunks);
    EW_REQUIRE(ew_cdiv(rows, rpb) == w.chunks || true, "unreachable");
    // chunks whose first row is past the end would leave their partial slot unwritten: rpb*(chunks-1) < rows always holds
    dim3 grid(w.chunks, n_slabs);
    const size_t lds = (size_t)PL * 2 * C_src * sizeof(float);
    EW_REQUIRE(lds <= 64 * 1024, "ew_groupnorm_stats_f16: C_src too large for the LDS combine");
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(VPP * PL), lds, (hipStream_t)stream, (const f16*)x, (const int8_t*)x_lo,
                       w.part, w.pivot, rows, C_src, c_off, C_tot, VPP, PL, rpb);
    return ew_check_launch("ew_groupnorm_stats_f16");
}

extern "C" ew_status ew_groupnorm_finalize(float* ws, int n_slabs, int rows, int C_tot, int groups, void* stream) {
    EW_REQUIRE(ws, "ew_groupnorm_finalize: null pointer");
    EW_REQUIRE(n_slabs > 0 && rows > 0 && groups > 0 && C_tot % groups == 0 && C_tot / groups <= 512,
               "ew_groupnorm_finalize: bad shape");
    const GnWs w = gn_ws(ws, n_slabs, rows, C_tot);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_slabs * groups), dim3(256), 0, (hipStream_t)stream,
                       w.part, w.pivot, w.stats, n_slabs, rows, C_tot, groups, w.chunks);
    return ew_check_launch("ew_groupnorm_finalize");
}

static ew_status gn_apply_launch(const char* what, const void* x, const void* x_lo, const float* ws, const void* gamma,
                                 const void* beta, void* y, void* y_lo, int ld_y, int n_slabs, int rows, int C_src, int c_off,
                                 int C_tot, int groups, float eps, int silu, void* stream) {
    EW_REQUIRE(x && ws && gamma && beta && y, "ew_groupnorm_apply: null pointer");
    EW_REQUIRE(n_slabs > 0 && rows > 0 && C_src > 0 && C_src % 8 == 0 && c_off % 8 == 0 && C_tot % 8 == 0,
               "ew_groupnorm_apply: bad shape");
    EW_REQUIRE(groups > 0 && C_tot % groups == 0 && c_off >= 0 && c_off + C_src <= C_tot, "ew_groupnorm_apply: bad groups");
    EW_REQUIRE(ld_y >= C_tot && ld_y % 8 == 0, "ew_groupnorm_apply: bad output row stride");
    const int VPP = C_src / 8;
    EW_REQUIRE(VPP <= 1024, "ew_groupnorm_apply: C_src too large");
    const int PL = VPP >= 256 ? 1 : 256 / VPP;
    const int rpb = 8 * PL;                                    // rows per block: 8 per thread column (round 4: 32 -> 8, level-0 apply 139 -> 127 us: smaller blocks, better balance)
    const GnWs w = gn_ws((float*)ws, n_slabs, rows, C_tot);
    dim3 grid(ew_cdiv(rows, rpb), n_slabs);
#define GN_APPLY(LO_, YLO_)                                                                                                      \
    hipLaunchKernelGGL((gn_apply_kernel<LO_, YLO_>), grid, dim3(VPP * PL), 0, (hipStream_t)stream, (const f16*)x,                  \
                       (const int8_t*)x_lo, w.stats, (const f16*)gamma, (const f16*)beta, (f16*)y, (f16*)y_lo, ld_y, rows, C_src,  \
                       c_off, C_tot, groups, eps, silu, VPP, PL, rpb)
    if (x_lo) { if (y_lo) GN_APPLY(true, true); else GN_APPLY(true, false); }
    else      { if (y_lo) GN_APPLY(false, true); else GN_APPLY(false, false); }
#undef GN_APPLY
    return ew_check_launch(what);
}

extern "C" ew_status ew_groupnorm_apply_f16(const void* x, const void* x_lo, const float* ws, const void* gamma,
                                            const void* beta, void* y, int n_slabs, int rows, int C_src, int c_off, int C_tot,
                                            int groups, float eps, int silu, void* stream) {
    return gn_apply_launch("ew_groupnorm_apply_f16", x, x_lo, ws, gamma, beta, y, nullptr, C_tot, n_slabs, rows, C_src, c_off, C_tot,
                           groups, eps, silu, stream);
}

extern "C" ew_status ew_groupnorm_apply_split_f16(const void* x, const void* x_lo, const float* ws, const void* gamma,
                                                  const void* beta, void* y, void* y_lo, int ld_y, int n_slabs, int rows, int C_src,
                                                  int c_off, int C_tot, int groups, float eps, int silu, void* stream) {
    EW_REQUIRE(y_lo, "ew_groupnorm_apply_split_f16: null pointer");
    return gn_apply_launch("ew_groupnorm_apply_split_f16", x, x_lo, ws, gamma, beta, y, y_lo, ld_y, n_slabs, rows, C_src, c_off, C_tot,
                           groups, eps, silu, stream);
}

extern "C" ew_status ew_layernorm_f16(const void* x, const void* x_lo, const void* addvec, int rows_per_group, void* x_out,
                                      void* x_out_lo, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                                      void* stream) {
    EW_REQUIRE(x && gamma && beta && y, "ew_layernorm_f16: null pointer");
    EW_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 2048, "ew_layernorm_f16: need C %% 8 == 0 and C <= 2048 (C=%d)", C);
    EW_REQUIRE(!addvec || rows_per_group >= 1, "ew_layernorm_f16: rows_per_group must be >= 1");
    EW_REQUIRE(!x_out_lo || (addvec && x_out), "ew_layernorm_f16: x_out_lo needs addvec and x_out");
    const int nv = (C / 8 + 63) / 64;
    dim3 block(256);
    hipStream_t s = (hipStream_t)stream;
    const int rpg = rows_per_group >= 1 ? rows_per_group : 1;
#define LN_LAUNCH(NV, RPW)                                                                                         \
    hipLaunchKernelGGL((ln_kernel<NV, RPW>), dim3(ew_cdiv(rows, 4 * RPW)), block, 0, s, (const f16*)x, (const int8_t*)x_lo, \
                       (const f16*)addvec, rpg, (f16*)x_out, (int8_t*)x_out_lo, (const f16*)gamma, (const f16*)beta, (f16*)y, rows, C, eps)
    if (nv == 1) LN_LAUNCH(1, 4);
    else if (nv == 2) LN_LAUNCH(2, 4);
    else if (nv == 3) LN_LAUNCH(3, 2);
    else LN_LAUNCH(4, 2);
#undef LN_LAUNCH
    return ew_check_launch("ew_layernorm_f16");
}
