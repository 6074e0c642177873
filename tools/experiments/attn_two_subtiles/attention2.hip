// attention2.hip -- spatial self-attention, log2 form, TWO 32-query sub-tiles per wave (round 6 experiment -> see DESIGN.md 3.6 for the verdict).
//
// Same algorithm, fragment layouts and LDS tile format as attn_spatial_kernel<true> (attention.hip: swapped product S^T = K Q^T on MFMA 32x32x16,
// pre-scaled q / k with the running max subtracted through the MFMA's C operand, lazy max, P in registers, V^T tile in the accumulator's key order),
// but a wave owns 64 queries = two sub-tiles that SHARE every K / V^T fragment read: one ds_read_b128 feeds two MFMAs, and the per-tile fixed
// costs of a wave (its quarter of the K / V tile loads and LDS stores, the barrier, the loop: 1160 of the 3596 clocks of the round-5 cycle
// anatomy) are paid once per 32 MFMAs instead of once per 16.  A workgroup = 4 waves = 256 queries.
// The scores of a tile are processed one 32-key block at a time (QK^T of both sub-tiles -> exponentials -> packed P), so only 32 score registers
// are live; the rare max raise recomputes the tile's QK^T instead of keeping all 64 (first tile of a sequence, or a row sum beyond 2^10).
#include "common.h"
#include <type_traits>

namespace {

__device__ __forceinline__ int swz2(int row) { return (row ^ (row >> 3)) & 7; }
typedef __fp16 h2x_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void attn_spatial2_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ vt,
                                                               f16* __restrict__ o, int S, int heads, int ld_qk, long long ld_vt, int ld_o,
                                                               int n_qtiles) {
    __shared__ __attribute__((aligned(16))) char smem[32768];  // 2 x { K tile [64 keys][64 d] | V^T tile [64 d][64 keys] }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware remap: all q-tiles of a (sequence, head) pair run on one XCD so K/V stay in that L2
        const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    }
    const int pair = bid / n_qtiles, qtile = bid - pair * n_qtiles;
    const int seq = pair / heads, head = pair - seq * heads;
    const long long tok0 = (long long)seq * S;

    // Q fragments (B operand) of the two sub-tiles: query = q0 + 32 b + lq, d = 16 s + 8 lh + e
    f16x8 qf[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qi = qtile * 256 + wave * 64 + b * 32 + lq;
        const int q_ld = qi < S ? qi : S - 1;
        const f16* qp = q + (tok0 + q_ld) * ld_qk + head * 64 + lh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[b][s] = *(const f16x8*)(qp + s * 16);
    }

    // staging geometry (identical to attention.hip)
    const int srow = tid >> 2, sc = tid & 3;
    const f16* kbase = k + head * 64 + sc * 16;
    const f16* vbase = vt + (long long)(head * 64 + srow) * ld_vt + tok0 + sc * 16;
    const int k_w0 = srow * 128 + (((sc * 2) ^ swz2(srow)) << 4), k_w1 = srow * 128 + (((sc * 2 + 1) ^ swz2(srow)) << 4);
    f16x8 kr0, kr1, vr0, vr1;
    auto load_tile = [&](int key0) __attribute__((always_inline)) {
        int kr = key0 + srow;
        kr = kr < S ? kr : S - 1;
        const f16* kp = kbase + (tok0 + kr) * ld_qk;
        kr0 = *(const f16x8*)kp;
        kr1 = *(const f16x8*)(kp + 8);
        const int c = key0 + sc * 16;
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        vr0 = c < S ? *(const f16x8*)(vbase + key0) : z;
        vr1 = c + 8 < S ? *(const f16x8*)(vbase + key0 + 8) : z;
    };
    auto write_tile = [&](auto buf_tag) __attribute__((always_inline)) {
        char* const kl = smem + decltype(buf_tag)::value * 16384;
        *(f16x8*)(kl + k_w0) = kr0;
        *(f16x8*)(kl + k_w1) = kr1;
        typedef __attribute__((address_space(3))) char* lds_t;
        const unsigned a0 = (unsigned)(unsigned long long)(lds_t)(smem + k_w0), a1 = (unsigned)(unsigned long long)(lds_t)(smem + k_w1);
        constexpr int VOFF = decltype(buf_tag)::value * 16384 + 8192;
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t w0 = __builtin_bit_cast(u32x4_t, vr0), w1 = __builtin_bit_cast(u32x4_t, vr1);
        const u32x2_t v0l = {w0[0], w0[1]}, v0h = {w0[2], w0[3]}, v1l = {w1[0], w1[1]}, v1h = {w1[2], w1[3]};
        asm volatile("ds_write_b64 %0, %1 offset:%6\n\tds_write_b64 %0, %2 offset:%7\n\tds_write_b64 %3, %4 offset:%6\n\tds_write_b64 %3, %5 offset:%7"
                     :: "v"(a0), "v"(v0l), "v"(v1l), "v"(a1), "v"(v0h), "v"(v1h), "n"(VOFF), "n"(VOFF + 8) : "memory");
    };

    f32x16 oacc[2][2];
    f32x16 negm[2];                          // C operand of the first score MFMA of a block = -m_run of the sub-tile's query in every element
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[b][0][i] = 0.f; oacc[b][1][i] = 0.f; negm[b][i] = 0.f; }

    int foff[2][4];
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) foff[bq][sq] = (bq * 32 + lq) * 128 + (((2 * sq + lh) ^ swz2(bq * 32 + lq)) << 4);

    const int nt = (S + 63) / 64;
    load_tile(0);
    write_tile(std::integral_constant<int, 0>{});
    if (nt > 1) load_tile(64);
    __syncthreads();
    constexpr float LAZY_SUM_THR = 1024.f, LAZY_RAISE_THR = 4.0f;

    auto tile_step = [&](const int j, auto buf_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        const int key0 = j * 64;
        const char* kb = smem + BUF * 16384;
        const char* vb = kb + 8192;
        if (j + 1 < nt) write_tile(std::integral_constant<int, BUF ^ 1>{});
        if (j + 2 < nt) load_tile(key0 + 128);

        // scores of one 32-key block for both sub-tiles: S^T - m = K Q^T + (-m)
        auto scores = [&](const int blk, f32x16 (&sacc)[2]) __attribute__((always_inline)) {
            sacc[0] = negm[0];
            sacc[1] = negm[1];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f16x8 kf = *(const f16x8*)(kb + foff[blk][s]);
                sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0][s], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[1][s], sacc[1], 0, 0, 0);
            }
            if (key0 + 64 > S) {                 // ragged last tile
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= S) { sacc[0][r] = -INFINITY; sacc[1][r] = -INFINITY; }
                }
            }
        };
        // the rare path: tile max of both sub-tiles (QK^T recomputed block by block), raise, rescale
        auto raise = [&]() __attribute__((always_inline)) {
            float tmax[2] = {-INFINITY, -INFINITY};
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x16 sacc[2];
                scores(blk, sacc);
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) tmax[b] = fmaxf(fmaxf(tmax[b], sacc[b][r]), sacc[b][r + 1]);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float tm = fmaxf(tmax[b], __shfl_xor(tmax[b], 32, 64));
                const float delta = FIRST ? tm : (tm > LAZY_RAISE_THR ? tm : 0.f);
                if constexpr (!FIRST) {
                    const float alpha = exp2f(-delta);
                    l_run[b] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { oacc[b][0][i] *= alpha; oacc[b][1][i] *= alpha; }
                }
                m_run[b] += delta;
#pragma unroll
                for (int i = 0; i < 16; ++i) negm[b][i] -= delta;
            }
        };
        f16x8 pf[2][4];
        float lt[2];
        auto compute_p = [&]() __attribute__((always_inline)) {
            lt[0] = lt[1] = 0.f;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x16 sacc[2];
                scores(blk, sacc);
                __builtin_amdgcn_sched_barrier(0);      // keep the next block's QK^T below this block's exponentials: 32 score registers live, not 64
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        unsigned w[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = g2 * 8 + e * 2;
                            const h2x_t ph = __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(sacc[b][r]), __builtin_amdgcn_exp2f(sacc[b][r + 1]));
                            const h2x_t one = {(__fp16)1.0f, (__fp16)1.0f};
                            lt[b] = __builtin_amdgcn_fdot2(ph, one, lt[b], false);
                            w[e] = __builtin_bit_cast(unsigned, ph);
                        }
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 wv = {w[0], w[1], w[2], w[3]};
                        pf[b][blk * 2 + g2] = __builtin_bit_cast(f16x8, wv);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (FIRST) raise();
        compute_p();
        if constexpr (!FIRST) {
            if (__any(!(lt[0] <= LAZY_SUM_THR) || !(lt[1] <= LAZY_SUM_THR))) { raise(); compute_p(); }
        }
        l_run[0] += lt[0];
        l_run[1] += lt[1];
        // ---- O^T += V^T P^T : one V^T fragment read feeds both sub-tiles ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const f16x8 vf = *(const f16x8*)(vb + foff[db][g]);
                oacc[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[0][g], oacc[0][db], 0, 0, 0);
                oacc[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[1][g], oacc[1][db], 0, 0, 0);
            }
        }
        __syncthreads();
    };
    tile_step(0, std::integral_constant<int, 0>{}, std::true_type{});
    int j = 1;
    for (; j + 1 < nt; j += 2) {
        tile_step(j, std::integral_constant<int, 1>{}, std::false_type{});
        tile_step(j + 1, std::integral_constant<int, 0>{}, std::false_type{});
    }
    if (j < nt) tile_step(j, std::integral_constant<int, 1>{}, std::false_type{});

    // ---- normalise + store: lane holds query qi, d = 32 db + 8 (r >> 2) + 4 lh + (r & 3) ----
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qi = qtile * 256 + wave * 64 + b * 32 + lq;
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        const float inv = 1.0f / l_tot;
        if (qi < S) {
            f16* op = o + (tok0 + qi) * ld_o + head * 64 + 4 * lh;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f16x4 v = {(f16)(oacc[b][db][rq * 4 + 0] * inv), (f16)(oacc[b][db][rq * 4 + 1] * inv),
                                     (f16)(oacc[b][db][rq * 4 + 2] * inv), (f16)(oacc[b][db][rq * 4 + 3] * inv)};
                    *(f16x4*)(op + db * 32 + rq * 8) = v;
                }
        }
    }
}

}  // namespace

ew_status ew_attn_spatial2_launch(const void* q, const void* k, const void* vt, void* o, int n_seq, int S, int heads, int ld_qk, long long ld_vt,
                                  int ld_o, void* stream) {
    const int n_qtiles = ew_cdiv(S, 256);
    const long long nblk = (long long)n_seq * heads * n_qtiles;
    hipLaunchKernelGGL(attn_spatial2_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const f16*)q, (const f16*)k, (const f16*)vt,
                       (f16*)o, S, heads, ld_qk, ld_vt, ld_o, n_qtiles);
    return ew_check_launch("ew_attn_spatial_log2_f16");
}
