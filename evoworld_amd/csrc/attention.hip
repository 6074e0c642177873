// attention.hip -- self-attention cores of the spatio-temporal transformer blocks (gfx950, head_dim 64).
//
//  * ew_attn_spatial_f16: flash-style tiled softmax(QK^T)V over S = H*W tokens per frame (S up to 9216 at
//    config 2, 32768 at config 5).  MFMA 32x32x16 f16.  The score tile is computed SWAPPED, S^T = K Q^T, so
//    that every lane owns ONE query column: the online-softmax max/sum are lane-local (one cross-half
//    exchange per tile) and the O^T accumulator rescale is a per-lane scalar.  P feeds the second MFMA
//    straight from registers: the key order inside each 16-key MFMA step is whatever the S^T fragment
//    layout yields, and the V^T tile is written to LDS in that same order, so no cross-lane shuffle of P
//    is needed.  V arrives TRANSPOSED from the projection GEMM (swapped-operand ew_gemm_f16), so the PV
//    A-operand is a plain ds_read_b128.
//  * ew_attn_temporal_f16: attention over the frame axis (T<=32) for every (batch, pixel, head).  It is
//    HBM-bound (0.13 TFLOP per forward vs ~1.2 GB of q/k/v per call at level 0), so it runs on the VALU
//    (v_dot2 for QK^T, fp32 FMA for PV) with K/V rows broadcast from LDS; the [B*T,S,C]<->[B*S,T,C] regroup
//    of the reference is pure addressing.
// Reference call sites: diffusers AttnProcessor2_0 / F.scaled_dot_product_attention inside
// BasicTransformerBlock.attn1 and TemporalBasicTransformerBlock.attn1, instantiated through
// evoworld/trainer/unet_plucker.py:13,161-233.
#include "common.h"
#include <type_traits>

// Softmax scale-and-shift as two plain v_fma_f32 instead of one v_pk_fma_f32 (round 3): at SIMD level a packed fp32 instruction costs two
// plain issues anyway, and beside MFMAs it costs MORE than that (tools/experiments/mb_mfma_valu.hip: 8 MFMA + 32 v_pk_fma 308 ns,
// 8 MFMA + 64 v_fma 255 ns per iteration).  Bit-identical results, 840 -> 855 TF/s at the level-0 shape.  The file is compiled with
// -fno-slp-vectorize so that hipcc does not re-pack them.
namespace {

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }
typedef __fp16 h2_t __attribute__((ext_vector_type(2)));

// PRE = true (round 4, ew_attn_spatial_log2_f16): q and k arrive PRE-SCALED -- the projection GEMM's epilogue multiplied both by
// sqrt(scale * log2 e) in fp32 before its single rounding to fp16 -- so q.k is already the exponent in log2 units, and the running max is
// subtracted by the MFMA itself: the first MFMA of a score block takes C = (-m, ..., -m) (a 16-register tuple that only changes when the
// deferred max is raised) instead of C = 0.  The per-score v_fma_f32 (32 of the ~120 VALU instructions of a 64-key tile in this VALU-bound
// loop) disappears; a raise (rare) subtracts the increment from the tile's scores afterwards.
template <bool PRE>
__global__ __launch_bounds__(256, PRE ? 3 : 2) void attn_spatial_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                               const f16* __restrict__ vt, f16* __restrict__ o, int S,
                                                               int heads, int ld_qk, long long ld_vt, int ld_o, float sl2,
                                                               int n_qtiles) {
    __shared__ __attribute__((aligned(16))) char smem[32768];  // 2 x { K tile [64 keys][64 d] | V^T tile [64 d][64 keys] }
    char* const kl = smem;
    char* const vl = smem + 8192;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;

    // XCD-aware remap: all q-tiles of a (sequence, head) pair run on one XCD so K/V stay in that L2
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    }
    const int pair = bid / n_qtiles, qtile = bid - pair * n_qtiles;
    const int seq = pair / heads, head = pair - seq * heads;
    const long long tok0 = (long long)seq * S;

    // Q fragments (B operand): query = q0 + lq, d = 16*s + 8*lh + e
    const int q_idx = qtile * 128 + wave * 32 + lq;
    const int q_ld = q_idx < S ? q_idx : S - 1;
    f16x8 qf[4];
    {
        const f16* qp = q + (tok0 + q_ld) * ld_qk + head * 64 + lh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const f16x8*)(qp + s * 16);
    }

    // staging geometry
    const int srow = tid >> 2, sc = tid & 3;
    const f16* kbase = k + head * 64 + sc * 16;
    const f16* vbase = vt + (long long)(head * 64 + srow) * ld_vt + tok0 + sc * 16;
    const int k_w0 = srow * 128 + (((sc * 2) ^ swz(srow)) << 4), k_w1 = srow * 128 + (((sc * 2 + 1) ^ swz(srow)) << 4);

    f16x8 kr0, kr1, vr0, vr1;
    auto load_tile = [&](int key0) {
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
    // Full tiles (all but a ragged last one) load through per-lane pointers advanced by one tile per iteration: the generic
    // form above costs a 64-bit multiply-add and two clamps per tile and lane, and this loop is VALU-bound.
    const f16* kp_run = kbase + (tok0 + srow) * ld_qk;      // tile 0 (only dereferenced for tiles that lie completely below S)
    const f16* vp_run = vbase;
    auto load_tile_full = [&]() __attribute__((always_inline)) {
        kr0 = *(const f16x8*)kp_run;
        kr1 = *(const f16x8*)(kp_run + 8);
        vr0 = *(const f16x8*)vp_run;
        vr1 = *(const f16x8*)(vp_run + 8);
    };
    auto write_tile = [&](auto buf_tag) __attribute__((always_inline)) {
        char* const kl = smem + decltype(buf_tag)::value * 16384;     // compile-time buffer: the offset folds into the ds_write
        char* const vl = kl + 8192;
        *(f16x8*)(kl + k_w0) = kr0;
        *(f16x8*)(kl + k_w1) = kr1;
        // 16-key group -> slot A = keys {0..3, 8..11}, slot B = keys {4..7, 12..15}
        // the loop is VALU-bound: the repack into two 16-byte registers cost eight v_mov per tile; four 8-byte LDS writes cost none
        // (inline asm: hipcc merges adjacent 8-byte LDS stores back into ds_write_b128 + the moves; the __syncthreads after every write_tile waits
        // with lgkmcnt(0) whatever the compiler tracked)
        typedef __attribute__((address_space(3))) char* lds_t;
        const unsigned a0 = (unsigned)(unsigned long long)(lds_t)(smem + k_w0), a1 = (unsigned)(unsigned long long)(lds_t)(smem + k_w1);
        constexpr int VOFF = decltype(buf_tag)::value * 16384 + 8192;       // buffer and V^T half as the instruction's immediate offset: one address pair for both buffers
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t w0 = __builtin_bit_cast(u32x4_t, vr0), w1 = __builtin_bit_cast(u32x4_t, vr1);
        const u32x2_t v0l = {w0[0], w0[1]}, v0h = {w0[2], w0[3]}, v1l = {w1[0], w1[1]}, v1h = {w1[2], w1[3]};
        asm volatile("ds_write_b64 %0, %1 offset:%6\n\tds_write_b64 %0, %2 offset:%7\n\tds_write_b64 %3, %4 offset:%6\n\tds_write_b64 %3, %5 offset:%7"
                     :: "v"(a0), "v"(v0l), "v"(v1l), "v"(a1), "v"(v0h), "v"(v1h), "n"(VOFF), "n"(VOFF + 8) : "memory");
    };

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }
    float m_run = PRE ? 0.f : -INFINITY, l_run = 0.f;
    f32x16 negm;                            // PRE: C operand of the first score MFMA = -m_run in every element (0 until the first tile's raise)
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;

    // fragment byte offsets inside a tile, shared by the K tile (row = key) and the V^T tile (row = d): row-block b (0/1),
    // 16-byte slot pair s -> (b*32+lq)*128 + (((2s+lh) ^ swz(row)) << 4).  Precomputed: the loop was VALU-bound
    // (rocprofv3: 23 VALU instructions per MFMA, VALU busy ~90 %), address arithmetic included.
    int foff[2][4];
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) foff[bq][sq] = (bq * 32 + lq) * 128 + (((2 * sq + lh) ^ swz(bq * 32 + lq)) << 4);

    constexpr float DEFER_THR = 6.0f;   // log2 units: the running max is raised only when a tile exceeds it by 2^6 (T13)
    const int nt = (S + 63) / 64;
    load_tile(0);
    write_tile(std::integral_constant<int, 0>{});
    if (nt > 1) {                       // tile 1 waits in the registers; tile_step(0) stores it
        kp_run += (long long)64 * ld_qk;
        vp_run += 64;
        if (128 <= S) load_tile_full(); else load_tile(64);
    }
    __syncthreads();
    // One 64-key tile; the LDS buffer it reads (BUF) and the one it refills (BUF ^ 1) are compile-time constants: the loop is
    // unrolled by two below, so buffer selection costs no VALU (it used to be eight xors on the fragment offsets plus address
    // arithmetic on the four tile stores per iteration).
    constexpr bool LAZY = PRE;
    constexpr float LAZY_SUM_THR = 1024.f;   // a lane's 32 exponentials of a tile may sum to 2^10 before the max is looked at (each <= 2^10: far inside fp16)
    constexpr float LAZY_RAISE_THR = 4.0f;   // ... and then every query whose tile max exceeds the running max by 2^4 is raised (a lane over the sum limit holds
                                             // an element >= 2^5, so its query always is)
    auto tile_step = [&](const int j, auto buf_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;        // LAZY: the sequence's first tile (peeled: it always sets the running max)
        const int key0 = j * 64;
        const char* kb = kl + BUF * 16384;
        const char* vb = vl + BUF * 16384;
        // the other buffer was last read in iteration j-1 and every wave has passed that barrier: tile j+1 (requested a whole tile ago) goes in now,
        // and the registers are free for tile j+2
        if (j + 1 < nt) write_tile(std::integral_constant<int, BUF ^ 1>{});
        if (j + 2 < nt) {
            kp_run += (long long)64 * ld_qk;
            vp_run += 64;
            if (key0 + 192 <= S) load_tile_full(); else load_tile(key0 + 128);
        }

        // ---- S^T = K Q^T : two 32-key blocks ----
        f32x16 sacc[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            if constexpr (PRE) {
                sacc[blk] = negm;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[blk][i] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f16x8 kf = *(const f16x8*)(kb + foff[blk][s]);
                sacc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[blk], 0, 0, 0);
            }
        }
        // ---- mask the ragged last tile ----
        if (key0 + 64 > S) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= S) sacc[blk][r] = -INFINITY;
                }
        }
        // ---- online softmax (this lane: one query, 32 of the 64 keys; partner lane^32 has the rest) ----
        // LAZY: the max chain + cross-half exchange + rescale as a function, run on the first tile and when a row sum says so
        auto lazy_fix = [&]() __attribute__((always_inline)) {
            float tmax = fmaxf(sacc[0][0], sacc[0][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[0][r + 1]);
#pragma unroll
            for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[1][r]), sacc[1][r + 1]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            // the scores are already s - m_run: tmax is the increment.  First tile: it sets the max whatever its sign (m_run starts at 0, not -inf:
            // an infinite C operand would poison the MFMA) and nothing is accumulated yet.
            const float delta = FIRST ? tmax : (tmax > LAZY_RAISE_THR ? tmax : 0.f);
            if constexpr (!FIRST) {
                const float alpha = exp2f(-delta);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
            }
            m_run += delta;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sacc[0][i] -= delta; sacc[1][i] -= delta; negm[i] -= delta; }       // (negm in place: -(m + d) == (-m) - d exactly)
        };
        if constexpr (LAZY) {
            if constexpr (FIRST) lazy_fix();
        } else {
        float tmax = fmaxf(sacc[0][0], sacc[0][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[0][r + 1]);      // v_max3_f32 chain
#pragma unroll
        for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[1][r]), sacc[1][r + 1]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        // deferred max: raise the running max only when this tile exceeds it by > 2^DEFER_THR (P stays <= 2^6, exact in
        // fp16's relative precision); the O / l rescale (34 VALU per lane) is skipped on almost every tile.
        if constexpr (PRE) {
            // the scores are already s - m_run: tmax is the increment.  The first tile always sets the max (m_run starts at 0, not -inf:
            // an infinite C operand would poison the MFMA), whatever its sign.
            const bool raise = tmax > (j == 0 ? -INFINITY : DEFER_THR);
            if (__any(raise)) {
                const float delta = raise ? tmax : 0.f;
                const float alpha = j == 0 ? 1.f : exp2f(-delta);           // nothing accumulated yet on the first tile (exp2(-delta) may be inf)
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
#pragma unroll
                for (int i = 0; i < 16; ++i) { sacc[0][i] -= delta; sacc[1][i] -= delta; negm[i] = -m_run; }
            }
        } else {
        const float mt = tmax * sl2;
        const bool raise = mt > m_run + DEFER_THR;
        if (__any(raise)) {
            const float m_new = raise ? mt : m_run;
            const float alpha = exp2f(m_run - m_new);       // 1 for the lanes that keep their max; 0 on the first tile
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
        }
        }
        }
        // P = exp2(S*c - m) -> fp16 pairs (round-toward-zero pack: one instruction per pair; its bias cancels because the
        // normaliser l below is accumulated from the SAME rounded values, with v_dot2)
        f16x8 pf[4];
        float lt = 0.f;                       // LAZY: this tile's row sum (32 of the 64 keys)
        auto compute_p = [&]() __attribute__((always_inline)) {
        if constexpr (LAZY) lt = 0.f;
        const f32x2 nm2 = {-m_run, -m_run}, sl22 = {sl2, sl2};
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                unsigned w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = g2 * 8 + e * 2;
                    const float t[2] = {PRE ? sacc[blk][r] : fmaf(sacc[blk][r], sl2, nm2[0]), PRE ? sacc[blk][r + 1] : fmaf(sacc[blk][r + 1], sl2, nm2[0])};
                    const h2_t ph = __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]));
                    const h2_t one = {(__fp16)1.0f, (__fp16)1.0f};
                    if constexpr (LAZY) lt = __builtin_amdgcn_fdot2(ph, one, lt, false);
                    else l_run = __builtin_amdgcn_fdot2(ph, one, l_run, false);
                    w[e] = __builtin_bit_cast(unsigned, ph);
                }
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 wv = {w[0], w[1], w[2], w[3]};
                pf[blk * 2 + g2] = __builtin_bit_cast(f16x8, wv);
            }
        };
        compute_p();
        if constexpr (LAZY) {
            if constexpr (!FIRST) {
                // !(lt <= thr) also catches a NaN sum (inf - inf cannot occur here, but an overflowed exponential must never pass)
                if (__any(!(lt <= LAZY_SUM_THR))) { lazy_fix(); compute_p(); }
            }
            l_run += lt;
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {       // 16-key group (MFMA k-step)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const f16x8 vf = *(const f16x8*)(vb + foff[db][g]);
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[g], oacc[db], 0, 0, 0);
            }
        }
        __syncthreads();
    };
    if constexpr (LAZY) {
        tile_step(0, std::integral_constant<int, 0>{}, std::true_type{});
        int j = 1;
        for (; j + 1 < nt; j += 2) {
            tile_step(j, std::integral_constant<int, 1>{}, std::false_type{});
            tile_step(j + 1, std::integral_constant<int, 0>{}, std::false_type{});
        }
        if (j < nt) tile_step(j, std::integral_constant<int, 1>{}, std::false_type{});
    } else {
        int j = 0;
        for (; j + 1 < nt; j += 2) {
            tile_step(j, std::integral_constant<int, 0>{}, std::false_type{});
            tile_step(j + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
        if (j < nt) tile_step(j, std::integral_constant<int, 0>{}, std::false_type{});
    }
    // ---- normalise + store: lane holds query q_idx, d = 32*db + 8*(r>>2) + 4*lh + (r&3) ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_idx < S) {
        f16* op = o + (tok0 + q_idx) * ld_o + head * 64 + 4 * lh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f16x4 v = {(f16)(oacc[db][rq * 4 + 0] * inv), (f16)(oacc[db][rq * 4 + 1] * inv),
                                 (f16)(oacc[db][rq * 4 + 2] * inv), (f16)(oacc[db][rq * 4 + 3] * inv)};
                *(f16x4*)(op + db * 32 + rq * 8) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// temporal attention: one wave per (batch, pixel, head) problem; T <= 32 frames, head_dim 64.
// A problem's q / k / v / o are T rows of 128 bytes, S*ld apart in memory.  All global traffic is COALESCED (8 lanes per
// 128-byte row) and staged through ONE wave-private LDS region of 32 rows x 144 B that holds, in turn, q, k, v and o
// (144-byte rows: a lane reading its own row with ds_read_b128 is bank-conflict free).  The arithmetic is 8 MFMA 32x32x16:
//   S^T = K Q^T   (4 MFMA; lane = query t, 16 of the 32 keys each -> lane-local softmax + one cross-half exchange)
//   O^T = V^T P^T (4 MFMA; P stays in registers, the key order of the MFMA k-dimension follows the accumulator layout and
//                  V^T fragments are gathered from the v rows with 16-bit LDS reads in the same order)
// History: v1 had lane t load its own rows with eight 16-byte loads (32 lines per load instruction) and did the math on the
// VALU with 16 KB of LDS per wave: 1.9 TB/s; coalescing alone gave 2.2 TB/s (VALU-bound); this version is HBM-bound.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_temporal_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                            const f16* __restrict__ v, f16* __restrict__ o, int B, int T,
                                                            int S, int heads, int ld, int ld_o, float scale_log2e,
                                                            long long n_prob) {
    __shared__ __attribute__((aligned(16))) char smem_t[4 * 32 * 144];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lh = lane >> 5, lr = lane & 31;
    const long long pid = (long long)blockIdx.x * 4 + wave;
    const bool active = pid < n_prob;
    const long long pc = active ? pid : n_prob - 1;
    const int h = (int)(pc % heads);
    const long long bs = pc / heads;
    const int b = (int)(bs / S), s = (int)(bs - (long long)b * S);
    char* const reg = smem_t + wave * (32 * 144);
    const long long row0 = (long long)b * T * S + s;                         // row of frame 0; frame r is r*S rows further
    const int n_chunk = T * 8;

    // ---- all global loads first: chunk idx = it*64 + lane -> (row idx>>3, 16-byte column idx&7)
    f16x8 tq[4], tk[4], tv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane;
        const f16x8 z = {};
        tq[it] = z; tk[it] = z; tv[it] = z;
        if (idx < n_chunk) {
            const long long off = (row0 + (long long)(idx >> 3) * S) * ld + h * 64 + (idx & 7) * 8;
            tq[it] = *(const f16x8*)(q + off);
            tk[it] = *(const f16x8*)(k + off);
            tv[it] = *(const f16x8*)(v + off);
        }
    }
    auto put = [&](const f16x8 (&tt)[4]) {          // rows >= T are zero-filled (0 * garbage must not become NaN in P V)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 64 + lane;
            *(f16x8*)(reg + (idx >> 3) * 144 + (idx & 7) * 16) = tt[it];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    };
    auto done = [&]() {                              // every lane has consumed the region: it may be overwritten
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    };
    // ---- operand fragments: lane (row lr, k-chunk lh) holds elements [s*16 + lh*8, +8) of its row
    f16x8 qf[4], kf[4];
    put(tq);
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const f16x8*)(reg + lr * 144 + st * 32 + lh * 16);
    done();
    put(tk);
#pragma unroll
    for (int st = 0; st < 4; ++st) kf[st] = *(const f16x8*)(reg + lr * 144 + st * 32 + lh * 16);
    done();
    put(tv);

    // ---- S^T = K Q^T: lane = query lr, keys (r&3) + 8*(r>>2) + 4*lh
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[st], qf[st], sacc, 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= T) sacc[r] = -INFINITY;
        mx = fmaxf(mx, sacc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = -mx * scale_log2e;
    float l = 0.f;
    f16x8 pf[2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float pe = __builtin_amdgcn_exp2f(fmaf(sacc[g2 * 8 + e], scale_log2e, nm));
            const f16 ph = (f16)pe;
            l += (float)ph;                                             // the normaliser sums the SAME rounded values
            pf[g2][e] = ph;
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;

    // ---- O^T = V^T P^T: A fragment (row d = db*32 + lr, k-slot (g2, lh, e)) = v[key(g2, lh, e)][d]
    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            f16x8 vf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = 16 * g2 + 4 * lh + (e & 3) + 8 * (e >> 2);
                vf[e] = *(const f16*)(reg + key * 144 + (db * 32 + lr) * 2);
            }
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[g2], oacc[db], 0, 0, 0);
        }
    }
    done();
    // ---- o rows into the region (lane = query lr; d = 32*db + 8*(r>>2) + 4*lh + (r&3)), then coalesced stores
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (f16)(oacc[db][rq * 4 + e] * inv);
            *(f16x4*)(reg + lr * 144 + (32 * db + 8 * rq + 4 * lh) * 2) = ov;
        }
    done();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane;
        if (idx < n_chunk && active)
            *(f16x8*)(o + (row0 + (long long)(idx >> 3) * S) * ld_o + h * 64 + (idx & 7) * 8) =
                *(const f16x8*)(reg + (idx >> 3) * 144 + (idx & 7) * 16);
    }
}

// Temporal attention for 32 < T <= 64 frames (BASELINE.json configs[4]: T = 49): same coalesced staging and MFMA layout as
// attn_temporal_kernel, with two 32-row blocks on the query and on the key axis.  One wave per (batch, pixel, head); q, k
// and v rows live in three wave-private LDS regions of 64 rows x 144 B (rows >= T zero-filled); the o rows of a query block
// overwrite its q rows.
__global__ __launch_bounds__(256) void attn_temporal64_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                              const f16* __restrict__ v, f16* __restrict__ o, int B, int T,
                                                              int S, int heads, int ld, int ld_o, float scale_log2e,
                                                              long long n_prob) {
    extern __shared__ __attribute__((aligned(16))) char smem_t64[];       // 4 waves x 3 regions x 64 x 144
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lh = lane >> 5, lr = lane & 31;
    const long long pid = (long long)blockIdx.x * 4 + wave;
    const bool active = pid < n_prob;
    const long long pc = active ? pid : n_prob - 1;
    const int h = (int)(pc % heads);
    const long long bs = pc / heads;
    const int b = (int)(bs / S), s = (int)(bs - (long long)b * S);
    char* const rq = smem_t64 + wave * (3 * 64 * 144);
    char* const rk = rq + 64 * 144;
    char* const rv = rk + 64 * 144;
    const long long row0 = (long long)b * T * S + s;
    const int n_chunk = T * 8;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane;                                     // (row idx>>3, 16-byte column idx&7), rows 0..63
        f16x8 a = {}, bb = {}, c = {};
        if (idx < n_chunk) {
            const long long off = (row0 + (long long)(idx >> 3) * S) * ld + h * 64 + (idx & 7) * 8;
            a = *(const f16x8*)(q + off);
            bb = *(const f16x8*)(k + off);
            c = *(const f16x8*)(v + off);
        }
        const int lo = (idx >> 3) * 144 + (idx & 7) * 16;
        *(f16x8*)(rq + lo) = a;
        *(f16x8*)(rk + lo) = bb;
        *(f16x8*)(rv + lo) = c;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int nqb = T > 32 ? 2 : 1;
    for (int qb = 0; qb < nqb; ++qb) {
        f16x8 qf[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) qf[st] = *(const f16x8*)(rq + (qb * 32 + lr) * 144 + st * 32 + lh * 16);
        f32x16 sacc[2];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const f16x8 kf = *(const f16x8*)(rk + (kb * 32 + lr) * 144 + st * 32 + lh * 16);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], sacc[kb], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (key >= T) sacc[kb][r] = -INFINITY;
                mx = fmaxf(mx, sacc[kb][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nm = -mx * scale_log2e;
        float l = 0.f;
        f16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const f16 ph = (f16)__builtin_amdgcn_exp2f(fmaf(sacc[kb][g2 * 8 + e], scale_log2e, nm));
                    l += (float)ph;
                    pf[kb][g2][e] = ph;
                }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        f32x16 oacc[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    f16x8 vf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int key = kb * 32 + 16 * g2 + 4 * lh + (e & 3) + 8 * (e >> 2);
                        vf[e] = *(const f16*)(rv + key * 144 + (db * 32 + lr) * 2);
                    }
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][g2], oacc[db], 0, 0, 0);
                }
        }
        // all lanes hold their q fragments in registers: the q rows of this block become its o rows
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (f16)(oacc[db][r4 * 4 + e] * inv);
                *(f16x4*)(rq + (qb * 32 + lr) * 144 + (32 * db + 8 * r4 + 4 * lh) * 2) = ov;
            }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane;
        if (idx < n_chunk && active)
            *(f16x8*)(o + (row0 + (long long)(idx >> 3) * S) * ld_o + h * 64 + (idx & 7) * 8) =
                *(const f16x8*)(rq + (idx >> 3) * 144 + (idx & 7) * 16);
    }
}

}  // namespace

static ew_status attn_spatial_launch(const void* q, const void* k, const void* vt, void* o, int n_seq, int S, int heads, int ld_qk,
                                     long long ld_vt, int ld_o, float scale, bool pre, void* stream, const char* name) {
    EW_REQUIRE(q && k && vt && o, "%s: null pointer", name);
    EW_REQUIRE(n_seq > 0 && S > 0 && heads > 0, "%s: bad shape", name);
    EW_REQUIRE(S % 8 == 0, "%s: S must be a multiple of 8 (S=%d)", name, S);
    EW_REQUIRE(ld_qk % 8 == 0 && ld_vt % 8 == 0 && ld_o % 4 == 0, "%s: strides must be 16-byte aligned", name);
    const int n_qtiles = ew_cdiv(S, 128);
    const long long nblk = (long long)n_seq * heads * n_qtiles;
    EW_REQUIRE(nblk < 0x7fffffffLL, "%s: grid too large", name);
    if (pre)
        hipLaunchKernelGGL(attn_spatial_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const f16*)q,
                           (const f16*)k, (const f16*)vt, (f16*)o, S, heads, ld_qk, ld_vt, ld_o, 1.0f, n_qtiles);
    else
        hipLaunchKernelGGL(attn_spatial_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const f16*)q,
                           (const f16*)k, (const f16*)vt, (f16*)o, S, heads, ld_qk, ld_vt, ld_o,
                           scale * 1.4426950408889634f, n_qtiles);
    return ew_check_launch(name);
}

extern "C" ew_status ew_attn_spatial_f16(const void* q, const void* k, const void* vt, void* o, int n_seq, int S, int heads,
                                         int ld_qk, long long ld_vt, int ld_o, float scale, void* stream) {
    return attn_spatial_launch(q, k, vt, o, n_seq, S, heads, ld_qk, ld_vt, ld_o, scale, false, stream, "ew_attn_spatial_f16");
}

// q and k pre-scaled by sqrt(scale * log2 e) each (the caller's projection GEMM: ew_gemm_args.c_acc): softmax_2(q k^T) v
extern "C" ew_status ew_attn_spatial_log2_f16(const void* q, const void* k, const void* vt, void* o, int n_seq, int S, int heads,
                                              int ld_qk, long long ld_vt, int ld_o, void* stream) {
    return attn_spatial_launch(q, k, vt, o, n_seq, S, heads, ld_qk, ld_vt, ld_o, 1.0f, true, stream, "ew_attn_spatial_log2_f16");
}

extern "C" ew_status ew_attn_temporal_f16(const void* q, const void* k, const void* v, void* o, int B, int T, int S,
                                          int heads, int ld, int ld_o, float scale, void* stream) {
    EW_REQUIRE(q && k && v && o, "ew_attn_temporal_f16: null pointer");
    EW_REQUIRE(B > 0 && S > 0 && heads > 0 && T > 0 && T <= 64, "ew_attn_temporal_f16: need 0 < T <= 64 (T=%d)", T);
    EW_REQUIRE(ld % 8 == 0 && ld_o % 8 == 0, "ew_attn_temporal_f16: strides must be 16-byte aligned");
    const long long n_prob = (long long)B * S * heads;
    const long long nblk = (n_prob + 3) / 4;
    EW_REQUIRE(nblk < 0x7fffffffLL, "ew_attn_temporal_f16: grid too large");
    if (T > 32) {
        const size_t lds = 4 * 3 * 64 * 144;                        // 110,592 B
        static std::atomic<unsigned long long> attr_mask{0};                   // per (kernel instantiation, device)
    if (ew_status st = ew_ensure_dynamic_lds((const void*)attn_temporal64_kernel, (int)lds, attr_mask)) return st;
        hipLaunchKernelGGL(attn_temporal64_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, (const f16*)q,
                           (const f16*)k, (const f16*)v, (f16*)o, B, T, S, heads, ld, ld_o, scale * 1.4426950408889634f, n_prob);
        return ew_check_launch("ew_attn_temporal_f16");
    }
    hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const f16*)q,
                       (const f16*)k, (const f16*)v, (f16*)o, B, T, S, heads, ld, ld_o, scale * 1.4426950408889634f, n_prob);
    return ew_check_launch("ew_attn_temporal_f16");
}
