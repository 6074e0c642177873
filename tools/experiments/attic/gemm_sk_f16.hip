// ATTIC (round 2, measured and rejected; not built): see DESIGN.md section 3.2.  Correct (bit-exact tests passed), but slower than
// generation 3 on its target shapes: GEGLU 460800x2560x320 1370 us vs 1090 us.  Ablations (same kernel, pieces compiled out):
// no MFMA 1061, no W DMA 1238, no in-loop epilogue slices 843, no slices + no DMA 770, LDS reads + barriers only 371 us.
// Two findings: (1) a 32x160 wave tile needs 12 fragment reads per 20 MFMAs (generation 3: 14 per 40): LDS reading alone costs
// as much as the MFMAs; (2) GEGLU at K = 320 needs ~0.8 VALU cycles per MFMA cycle, so slices issued between the MFMAs of the
// same wave do not hide: both pipes and the wave's issue slot are full.
// gemm_sk_f16.hip -- "short-K" generation of the fused MFMA GEMM for gfx950: K = 320 (five 64-wide K-tiles), dense A, N a
// multiple of 320 -- the level-0 transformer projections of the U-Net (M = 460800 tokens): the GEGLU feed-forward
// up-projection (N = 2560), the fused q|k projection (N = 640) and the temporal q|k|v projection (N = 960).
//
// Why (rocprofv3 + bench.py per-shape breakdown, round 2): on these shapes generation 3 spends as long in its epilogue as
// in its main loop (GEGLU 460800x2560x320: 18.5 us per 256x320 tile = 10.3 us main loop + 8.2 us epilogue, against 5.3 us
// of MFMA time): five K-tiles per output tile cannot amortise an epilogue that all eight waves enter together, during which
// the matrix pipe idles; and the 160 accumulator registers per wave leave no room to overlap it.  Here:
//   * tile 128 x 320, 8 wave64 as 4(M) x 2(N), wave tile 32 x 160 = 80 accumulator VGPRs -> TWO accumulator sets: while
//     set A accumulates output tile n+1, the epilogue of tile n (set B) is issued in slices BETWEEN the MFMAs of the main
//     loop (bias / GEGLU VALU work and the global stores ride under the matrix pipe); the roles swap every tile (the tile
//     body is instantiated for both parities, all register indices are compile-time);
//   * the A tile (128 rows x K) is RESIDENT in LDS (80 KB) for all N-tiles of its M-block: only W streams (40 KB per
//     K-tile, two stages = 80 KB; 160 KB in total), 7.7 B staged per kFLOP against 10.7 for a streamed 128-row tile;
//   * no LDS epilogue patch: the W rows are staged PERMUTED (per-lane source rows), so that the fragments of one lane hold 8
//     consecutive output columns -> bias loads and output stores are 16-byte accesses, 64 contiguous bytes per row per
//     instruction (GEGLU: value fragments 2p, 2p+1 and their gate fragments; the fifth value / gate fragment is an 8-byte
//     tail);
//   * one persistent workgroup per CU walks M-blocks; at an M-block boundary the pending epilogue is flushed while the next
//     A tile and first W K-tile are in flight.
// Same argument block as the other generations (gemm_common.h); epilogues: bias (+SiLU / GELU, c_acc) and GEGLU.
#include "gemm_common.h"
#include <type_traits>

extern char g_gemm_last_kernel[64];

namespace {

#define SK_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define SK_FENCE() asm volatile("" ::: "memory")
#define SK_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SK_PIN() __builtin_amdgcn_sched_barrier(0)

constexpr int BM = 128, BN = 320, BK = 64, NK = 5, NW = 8, WAVES_N = 2;
constexpr int WM = 32, WN = 160, FM = 2, FN = 10;
constexpr int A_CHUNK = BM * 128;                 // 16 KB: 128 rows x one 64-wide K chunk
constexpr int A_TOTAL = NK * A_CHUNK;             // 80 KB
constexpr int W_STAGE = BN * 128;                 // 40 KB
constexpr int GB = BN / 8 / NW;                   // 5 W pieces per wave per K-tile
constexpr int GA = BM / 8 / NW;                   // 2 A pieces per wave per K chunk
constexpr int NSTEP = 2 * FN;                     // 20 steps (k-half, W fragment) per K-tile, FM MFMAs each
constexpr int PD = 4, RING = 5;                   // W fragments are read PD steps ahead of their MFMAs: a step is only FM = 2
                                                  // MFMAs (32 cycles), so two steps ahead (generation 3's distance) exposed the
                                                  // LDS latency at every step
constexpr int BAR_STEP = NSTEP - 1 - PD;

template <int EPI>
__global__ __launch_bounds__(64 * NW, 2) void gemmsk_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool GEGLU = EPI & 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int G = gridDim.x;
    const int mblocks = p.tiles_m, tiles_n = p.tiles_n;
    if ((int)blockIdx.x >= mblocks) return;

    const int srow = lane >> 3;
    const int slot = (lane & 7) ^ srow;

    // ---- per-lane source rows of the 5 W pieces of this wave (relative to the N-tile's first staged row)
    int wrow[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int R = 8 * (wave + NW * j) + srow;             // LDS row inside the W stage
        const int wn_ = R / WN, rem = R - wn_ * WN;
        const int f = rem >> 4, r = rem & 15, fk = r >> 2, e = r & 3;
        int g;
        if constexpr (GEGLU) {
            // fragments 0..4 = values, 5..9 = gates of out column c (tile-local, 160 per tile): host layout = blocks of 32
            // staged rows [16 value | 16 gate] per 16 output columns
            const int gate = f >= 5 ? 1 : 0, v = f - 5 * gate;
            const int c = wn_ * 80 + (v < 4 ? (v >> 1) * 32 + fk * 8 + (v & 1) * 4 + e : 64 + fk * 4 + e);
            g = 32 * (c >> 4) + (c & 15) + 16 * gate;
        } else {
            g = wn_ * WN + (f >> 1) * 32 + fk * 8 + (f & 1) * 4 + e;          // pair (2q, 2q+1) = 8 consecutive columns
        }
        wrow[j] = g * p.K + slot * 8;                          // element offset inside the N-tile's W slab
    }

    // ---- fragment geometry
    const int frow = lane & 15, fks = lane >> 4, sw = frow & 7;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int so = ((kh * 4 + fks) ^ sw) << 4;
        a_rd[kh] = (wm * WM + frow) * 128 + so;
        b_rd[kh] = A_TOTAL + (wn * WN + frow) * 128 + so;
    }

    f32x4 accA[FM][FN], accB[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) { accA[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    f16x8 af[2][FM];
    f16x8 bfr[RING];

    // ---- W staging: one K-tile = GB pieces per wave
    const f16* w_tile = p.w;                       // W slab of the N-tile being staged
    int w_kt = 0;                                  // K-tile index being staged (0..NK-1)
    char* w_buf = smem + A_TOTAL;
    auto w_piece = [&](int j) __attribute__((always_inline)) {
#if defined(SK_ABLATE) && (SK_ABLATE & 4)
        asm volatile("" ::"v"(w_tile + wrow[j] + w_kt * BK));                                                         // ablation: no W DMA
#else
        glds16(w_tile + wrow[j] + w_kt * BK, w_buf + (wave + NW * j) * 1024);
#endif
    };

    // ---- epilogue of one finished accumulator set (slices), coordinates of the tile it belongs to
    int ep_m0 = 0, ep_tn = 0;                      // M-block first row, N-tile index of the PENDING tile
    bool ep_pending = false;
    const f16* bp = p.bias ? p.bias : p.zero_page;
    const int mbias = p.bias ? 1 : 0;

    // slice s of the pending epilogue on accumulator set `acc`: GEGLU: 6 slices (i = s / 3, part = s % 3); else 10 (i = s / 5, q = s % 5)
    constexpr int NSLICE = GEGLU ? 6 : 10;
    // A slice = fetch (its bias vectors: global loads) + apply (VALU + one store).  vmcnt is an IN-ORDER counter: waiting for
    // a load also waits for every older VMEM operation, so inside the main loop the fetches of a K-tile's two slices are
    // issued at its very first step, BEFORE the K-tile's DMA pieces -- the wait at apply time then covers only operations
    // older than that DMA (measured with the loads issued next to their use: every slice stalled until the W K-tile in
    // flight had landed, 12 us per N-tile instead of 4).
    struct SliceOps { f16x4 b0, b1, b2, b3; f16x8 bv; };
    auto ep_fetch = [&](SliceOps& so, auto s_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(s_tag)::value;
        int ln = tid & 63;
        asm volatile("" : "+v"(ln));               // opaque: keep the lane-derived constants out of the main loop's live set
        const int fk = ln >> 4;
        if constexpr (GEGLU) {
            constexpr int PART = S % 3;
            if constexpr (PART < 2) {
                const int c0 = wn * 80 + PART * 32 + fk * 8;                       // tile-local first out column of this lane
                const int bi = ep_tn * BN + 32 * (c0 >> 4) + (c0 & 15);            // staged index of its value bias (8 = 4 + 4)
                so.b0 = *(const f16x4*)(bp + (bi)*mbias); so.b1 = *(const f16x4*)(bp + (bi + 4) * mbias);
                so.b2 = *(const f16x4*)(bp + (bi + 16) * mbias); so.b3 = *(const f16x4*)(bp + (bi + 20) * mbias);
            } else {
                const int c0 = wn * 80 + 64 + fk * 4;
                const int bi = ep_tn * BN + 32 * (c0 >> 4) + (c0 & 15);
                so.b0 = *(const f16x4*)(bp + bi * mbias); so.b2 = *(const f16x4*)(bp + (bi + 16) * mbias);
            }
        } else {
            constexpr int Q = S % 5;
            so.bv = *(const f16x8*)(bp + (ep_tn * BN + wn * WN + Q * 32 + fk * 8) * mbias);
        }
    };
    auto ep_apply = [&](f32x4 (&acc)[FM][FN], const SliceOps& so, auto s_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(s_tag)::value;
        int ln = tid & 63;
        asm volatile("" : "+v"(ln));
        const int fr = ln & 15, fk = ln >> 4;
        auto f4 = [](f16x4 h) { return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; };
        if constexpr (GEGLU) {
            constexpr int I = S / 3, PART = S % 3;
            const int m = ep_m0 + wm * WM + I * 16 + fr;
            if constexpr (PART < 2) {
                const int c0 = wn * 80 + PART * 32 + fk * 8;
                const f32x4 v0 = acc[I][2 * PART] + f4(so.b0), v1 = acc[I][2 * PART + 1] + f4(so.b1);
                const f32x4 g0 = acc[I][5 + 2 * PART] + f4(so.b2), g1 = acc[I][5 + 2 * PART + 1] + f4(so.b3);
                const f32x2 o0 = ew_vgelu2((f32x2){v0[0], v0[1]}, (f32x2){g0[0], g0[1]}), o1 = ew_vgelu2((f32x2){v0[2], v0[3]}, (f32x2){g0[2], g0[3]});
                const f32x2 o2 = ew_vgelu2((f32x2){v1[0], v1[1]}, (f32x2){g1[0], g1[1]}), o3 = ew_vgelu2((f32x2){v1[2], v1[3]}, (f32x2){g1[2], g1[3]});
                const f16x8 o = {(f16)o0[0], (f16)o0[1], (f16)o1[0], (f16)o1[1], (f16)o2[0], (f16)o2[1], (f16)o3[0], (f16)o3[1]};
                acc[I][2 * PART] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[I][2 * PART + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[I][5 + 2 * PART] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[I][5 + 2 * PART + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (m < p.M) *(f16x8*)(p.out + (size_t)m * p.ld_out + ep_tn * (BN / 2) + c0) = o;
            } else {
                const int c0 = wn * 80 + 64 + fk * 4;
                const f32x4 v0 = acc[I][4] + f4(so.b0), g0 = acc[I][9] + f4(so.b2);
                const f32x2 o0 = ew_vgelu2((f32x2){v0[0], v0[1]}, (f32x2){g0[0], g0[1]}), o1 = ew_vgelu2((f32x2){v0[2], v0[3]}, (f32x2){g0[2], g0[3]});
                const f16x4 o = {(f16)o0[0], (f16)o0[1], (f16)o1[0], (f16)o1[1]};
                acc[I][4] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[I][9] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (m < p.M) *(f16x4*)(p.out + (size_t)m * p.ld_out + ep_tn * (BN / 2) + c0) = o;
            }
        } else {
            constexpr int I = S / 5, Q = S % 5;
            const int m = ep_m0 + wm * WM + I * 16 + fr;
            const int n = ep_tn * BN + wn * WN + Q * 32 + fk * 8;
            const f32x4 a0 = acc[I][2 * Q], a1 = acc[I][2 * Q + 1];
            acc[I][2 * Q] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[I][2 * Q + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float vv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = vv[e] + (float)so.bv[e];
                if (p.act == EW_ACT_SILU) x = ew_silu(x);
                else if (p.act == EW_ACT_GELU) x = ew_gelu(x);
                o[e] = (f16)(x * p.c_acc);
            }
            if (m < p.M) *(f16x8*)(p.out + (size_t)m * p.ld_out + n) = o;
        }
    };
    auto ep_slice = [&](f32x4 (&acc)[FM][FN], auto s_tag) __attribute__((always_inline)) {
        SliceOps so;
        ep_fetch(so, s_tag);
        ep_apply(acc, so, s_tag);
    };
    auto ep_flush = [&](f32x4 (&acc)[FM][FN]) __attribute__((always_inline)) {
        if (!ep_pending) return;
        // slices are pinned apart: scheduled together, their operand loads and GELU temporaries are all hoisted to the top and
        // the allocator runs out of registers next to the 160 accumulators (417 spilled VGPRs measured)
        ep_slice(acc, std::integral_constant<int, 0>{}); SK_PIN();
        ep_slice(acc, std::integral_constant<int, 1>{}); SK_PIN();
        ep_slice(acc, std::integral_constant<int, 2>{}); SK_PIN();
        ep_slice(acc, std::integral_constant<int, 3>{}); SK_PIN();
        ep_slice(acc, std::integral_constant<int, 4>{}); SK_PIN();
        ep_slice(acc, std::integral_constant<int, 5>{}); SK_PIN();
        if constexpr (NSLICE > 6) {
            ep_slice(acc, std::integral_constant<int, 6>{}); SK_PIN();
            ep_slice(acc, std::integral_constant<int, 7>{}); SK_PIN();
            ep_slice(acc, std::integral_constant<int, 8>{}); SK_PIN();
            ep_slice(acc, std::integral_constant<int, 9>{}); SK_PIN();
        }
        ep_pending = false;
    };

    // ---- one K-tile of an output tile (N-tile tn of the resident M-block): `cur` accumulates, `prev` is drained in slices.
    // KT is compile-time so that (K-tile, step) -> epilogue slice is static register indexing.
    int s_cur = 0;                                           // W stage holding the K-tile being consumed
    auto ktile = [&](f32x4 (&cur)[FM][FN], f32x4 (&prev)[FM][FN], auto kt_tag, int tn, bool last_tile_of_block, bool drain)
                     __attribute__((always_inline)) {
        constexpr int KT = decltype(kt_tag)::value;
        const char* wc = smem + s_cur * W_STAGE;                              // b_rd carries A_TOTAL
        const char* wnx = smem + (s_cur ^ 1) * W_STAGE;
        const char* ac = smem + KT * A_CHUNK;
        const bool pend = !(last_tile_of_block && KT == NK - 1);
        if (pend) {
            if constexpr (KT == NK - 1) { w_tile = p.w + (size_t)(tn + 1) * BN * p.K; w_kt = 0; }
            else { w_kt = KT + 1; }
            w_buf = smem + A_TOTAL + (s_cur ^ 1) * W_STAGE;
        }
        if constexpr (KT == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i) af[0][i] = *(const f16x8*)(ac + a_rd[0] + i * 2048);
#pragma unroll
            for (int j = 0; j < PD; ++j) bfr[j] = *(const f16x8*)(wc + b_rd[0] + j * 2048);
        }
        constexpr int S0 = KT * 2, S1 = KT * 2 + 1;                 // the two epilogue slices scheduled inside this K-tile
        constexpr bool HAS0 = S0 < NSLICE, HAS1 = S1 < NSLICE;
        SliceOps so0, so1;
        if (drain) {
            if constexpr (HAS0) ep_fetch(so0, std::integral_constant<int, HAS0 ? S0 : 0>{});
            if constexpr (HAS1) ep_fetch(so1, std::integral_constant<int, HAS1 ? S1 : 0>{});
        }
        SK_PIN();
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            const int kh = t / FN, j = t - kh * FN;
            if (t + PD < NSTEP) {
                const int kh2 = (t + PD) / FN, j2 = (t + PD) - kh2 * FN;
                bfr[(t + PD) % RING] = *(const f16x8*)(wc + b_rd[kh2] + j2 * 2048);
            }
            if (t < FM) af[1][t] = *(const f16x8*)(ac + a_rd[1] + t * 2048);
            if constexpr (KT + 1 < NK) {
                if (t >= NSTEP - PD) {
                    bfr[(t + PD) % RING] = *(const f16x8*)(wnx + b_rd[0] + (t + PD - NSTEP) * 2048);
                    if (t == NSTEP - 1) {
#pragma unroll
                        for (int i = 0; i < FM; ++i) af[0][i] = *(const f16x8*)(ac + A_CHUNK + a_rd[0] + i * 2048);
                    }
                }
            }
            if (t < GB) {
                if (pend) w_piece(t);
            }
            SK_PIN();
#pragma unroll
            for (int i = 0; i < FM; ++i)
#if defined(SK_ABLATE) && (SK_ABLATE & 1)
                asm volatile("" ::"v"(bfr[t % RING]), "v"(af[kh][i]));                                                 // ablation: no MFMA
#else
                cur[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[t % RING], af[kh][i], cur[i][j], 0, 0, 0);       // D[n][m]
#endif
            SK_PIN();
            // ---- the previous tile's epilogue under the matrix pipe: slice S0 after step 6, slice S1 after step 13 (operands
            //      were fetched before this K-tile's DMA pieces)
            if (t == 6) {
                if constexpr (HAS0) { if (drain) ep_apply(prev, so0, std::integral_constant<int, HAS0 ? S0 : 0>{}); }
                SK_PIN();
            }
            if (t == 13) {
                if constexpr (HAS1) { if (drain) ep_apply(prev, so1, std::integral_constant<int, HAS1 ? S1 : 0>{}); }
                SK_PIN();
            }
            if (t == BAR_STEP) {
                // every fragment read of this K-tile's W stage has been issued; the staged next K-tile must have landed.  The
                // slice stores were issued AFTER the DMA pieces: counted wait, they stay in flight (vmcnt counts in issue order)
                constexpr int NST = (HAS0 ? 1 : 0) + (HAS1 ? 1 : 0);
                if (drain) {
                    if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if constexpr (NST == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else SK_WAIT_VM0();
                } else {
                    SK_WAIT_VM0();
                }
                SK_WAIT_LGKM0();
                SK_FENCE();
                __builtin_amdgcn_s_barrier();
                SK_FENCE();
            }
        }
        s_cur ^= 1;
    };
    auto run_tile = [&](f32x4 (&cur)[FM][FN], f32x4 (&prev)[FM][FN], int m0, int tn, bool last) __attribute__((always_inline)) {
#if defined(SK_ABLATE) && (SK_ABLATE & 8)
        const bool drain = false;                                                                                      // ablation: no in-loop slices
#else
        const bool drain = ep_pending;
#endif
        ktile(cur, prev, std::integral_constant<int, 0>{}, tn, last, drain);
        ktile(cur, prev, std::integral_constant<int, 1>{}, tn, last, drain);
        ktile(cur, prev, std::integral_constant<int, 2>{}, tn, last, drain);
        ktile(cur, prev, std::integral_constant<int, 3>{}, tn, last, drain);
        ktile(cur, prev, std::integral_constant<int, 4>{}, tn, last, drain);
        // `prev` is fully drained (all slices were scheduled inside the five K-tiles); `cur` becomes the pending tile
        ep_m0 = m0; ep_tn = tn; ep_pending = true;
    };

    // Accumulator roles are STATIC: every M-block starts on set A with nothing pending, tiles alternate A / B (the N-tile loop
    // is unrolled by two), and the last tile of the M-block is flushed while the next M-block's A tile and first W K-tile are
    // in flight.  (A run-time parity carried across M-blocks made the allocator shuffle both 80-register sets at every
    // merge point: ~400 spilled VGPRs.)
    bool first = true;
    for (int mb = blockIdx.x; mb < mblocks; mb += G) {
        const int m0 = mb * BM;
        if (first) {
            // A tile: 5 chunks x 2 pieces per wave; W K-tile 0 of N-tile 0
#pragma unroll
            for (int c = 0; c < NK; ++c)
#pragma unroll
                for (int jj = 0; jj < GA; ++jj) {
                    int m = m0 + (wave + NW * jj) * 8 + srow;
                    m = m < p.M ? m : p.M - 1;
                    glds16(p.a + (size_t)m * p.lda + c * BK + slot * 8, smem + c * A_CHUNK + (wave + NW * jj) * 1024);
                }
            w_tile = p.w; w_kt = 0; w_buf = smem + A_TOTAL + s_cur * W_STAGE;
#pragma unroll
            for (int j = 0; j < GB; ++j) w_piece(j);
            SK_WAIT_VM0();
            SK_FENCE();
            __builtin_amdgcn_s_barrier();
            SK_FENCE();
            first = false;
        }
        for (int tn = 0; tn < tiles_n; tn += 2) {
            run_tile(accA, accB, m0, tn, tn == tiles_n - 1);
            if (tn + 1 < tiles_n) run_tile(accB, accA, m0, tn + 1, tn + 1 == tiles_n - 1);
        }
        // ---- M-block boundary: every wave is past its last read of the A tile and of both W stages (last BAR_STEP barrier;
        //      the steps after it touch registers only)
        SK_WAIT_LGKM0();
        SK_FENCE();
        __builtin_amdgcn_s_barrier();
        SK_FENCE();
        const int mbn = mb + G;
        if (mbn < mblocks) {
            const int m0n = mbn * BM;
#pragma unroll
            for (int c = 0; c < NK; ++c)
#pragma unroll
                for (int jj = 0; jj < GA; ++jj) {
                    int m = m0n + (wave + NW * jj) * 8 + srow;
                    m = m < p.M ? m : p.M - 1;
                    glds16(p.a + (size_t)m * p.lda + c * BK + slot * 8, smem + c * A_CHUNK + (wave + NW * jj) * 1024);
                }
            w_tile = p.w; w_kt = 0; w_buf = smem + A_TOTAL + s_cur * W_STAGE;
#pragma unroll
            for (int j = 0; j < GB; ++j) w_piece(j);
        }
        // the last tile's epilogue drains while that DMA is in flight
        if (tiles_n & 1) ep_flush(accA); else ep_flush(accB);
        SK_WAIT_VM0();
        SK_FENCE();
        __builtin_amdgcn_s_barrier();
        SK_FENCE();
    }
}

template <int EPI>
ew_status launch_sk(const GemmP& p, hipStream_t s) {
    GemmP q = p;
    q.tiles_m = ew_cdiv(p.M, BM);
    q.tiles_n = (EPI & 8) ? p.N / BN : p.N / BN;
    const size_t lds = A_TOTAL + 2 * W_STAGE;                 // 163,840 B
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemmsk_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ew_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return EW_ERR_HIP; }
        attr_set = true;
    }
    int grid = 256;
    if (q.tiles_m < grid) grid = q.tiles_m;
    snprintf(g_gemm_last_kernel, 64, "gemmsk_kernel<%d>", EPI);
    hipLaunchKernelGGL((gemmsk_kernel<EPI>), dim3(grid), dim3(64 * NW), lds, s, q);
    return ew_check_launch("ew_gemm_f16(short-K)");
}

}  // namespace

// dense, K = 320, N a multiple of 320, no row-bias / residual / split outputs, enough M-blocks to fill the chip
bool ew_gemm_sk_wants(const GemmP& p) {
    if (p.mode != EW_A_DENSE || p.K != NK * BK || p.c2 != 0 || p.N % BN != 0) return false;
    if (p.rowbias || p.r1 || p.r2 || p.r1_lo || p.r2_lo || p.out_lo) return false;
    if (p.N / BN < 2) return false;                          // a single N-tile gains nothing from the resident A tile
    return p.M >= 256 * BM;
}

ew_status ew_gemm_sk_dispatch(const GemmP& p, hipStream_t s) {
    if (p.act == EW_ACT_GEGLU) return launch_sk<8>(p, s);
    return launch_sk<0>(p, s);
}
