// gemm3_f16.hip -- third-generation fused MFMA GEMM / implicit-GEMM conv for gfx950: 256x320 output tile per workgroup.
//
// Why (tools/experiments/exp13..16): generation 2's dense GEMMs run exactly as fast with their MFMAs and
// ds_reads compiled out -- they are bound by the LDS-DMA feed (global_load_lds_dwordx4), which saturates at ~50-54 GB/s per
// CU (~13.5 TB/s over the chip) for L2-resident operand streams, independent of tile shape, ring depth and row stride.
// The only lever is bytes staged per flop.  Every channel count of the U-Net is a multiple of 320, so:
//   * tile 256 x 320 x 64: (256+320)*128 B = 72 KB per 10.5 MFLOP = 7.0 B/kFLOP  (gen 2: 256x160 -> 10.2, 128x256 -> 11.7);
//   * 8 wave64 as 4(M) x 2(N), wave tile 64 x 160 = 160 accumulator VGPRs; the 256-VGPR budget leaves no room for
//     double-buffered fragments of a whole half-step, so the W fragments STREAM through a 4-deep register ring (read two
//     steps ahead of the 4 MFMAs that consume them) and the A fragments of the next k-half are read during the current one;
//   * two LDS stages of 72 KB (double buffer): the DMA of K-tile v+1 is issued one piece per step over the first 9 of the 20
//     steps of K-tile v, right after the barrier that freed its slot; one barrier per K-tile, placed after the LAST fragment
//     read of the tile (step 17 of 20) so that steps 18-19 already prefetch the next tile's first fragments;
//   * persistent workgroups, XCD-chunked tile order, fused epilogue (bias / row-bias / SiLU / GEGLU / 2 residuals) through a
//     wave-private fp32 LDS patch in two 80-column passes, as in generation 2.
// Same argument block and A addressing modes (dense / conv3x3 / temporal 3-tap, dual source, zero page) as gen 1/2.
// Requires N % 320 == 0 (the dispatcher falls back to generation 2 otherwise).
#include "gemm_common.h"
#include <type_traits>
#include <mutex>

extern char g_gemm_last_kernel[64];

namespace {

#define EW3_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define EW3_FENCE() asm volatile("" ::: "memory")
#define EW3_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define EW3_LDS(ptr) (*(const f16x8*)(ptr))

// The file is compiled twice (Makefile): EW3_BN = 320 (every channel count of the U-Net) and EW3_BN = 256 (gemm3b_f16.o: the VAE's
// 256- / 512-channel convs, which otherwise run on generation 2's 128x256 tile at about a third of this kernel's rate).
#ifndef EW3_BN
#define EW3_BN 320
#endif
#if EW3_BN == 320
#define EW3_NAME(x) x
#define EW3_KERNEL_STR "gemm3_kernel"
#else
#define EW3_NAME(x) x##_b256
#define EW3_KERNEL_STR "gemm3b_kernel"
#define gemm3_kernel gemm3b_kernel
#endif
constexpr int BM = 256, BN = EW3_BN, BK = 64, NW = 8, WAVES_N = 2;
constexpr int WM = 64, WN = BN / WAVES_N, FM = 4, FN = WN / 16;
static_assert(BN % 64 == 0 && FN % 2 == 0 && (FM * FN) % 8 == 0, "tile geometry");
constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW, NP = GA + GB;        // 4 + 5 DMA pieces per wave per K-tile
constexpr int NSTEP = 2 * FN;                                          // 20 steps (k-half, W fragment) per K-tile
#define EW3_PIN() __builtin_amdgcn_sched_barrier(0)
constexpr int ITEMS_BYTES = 8192;                                      // work-item table: up to 512 (tile, k0, k1) entries per block
constexpr int PD = 2;                                         // W fragments are read PD steps ahead of their MFMAs (ring of 4)
constexpr int BAR_STEP = NSTEP - 1 - PD;                               // barrier after the step that issues the tile's last read

// (Round 5's halo-slab A loader for the stride-1 3x3 convs -- one slab per (channel chunk, kernel row) serving the three kernel columns as address
// shifts, bit-identical, measured +-0 in time and in FETCH_SIZE, profiles/r05_f_halo_slab_* -- was removed from the source in round 6: commit e42126f.)

// Tile id -> (tm, tn).  Ids are consumed in XCD-contiguous chunks of 32 (one per CU of an XCD at a time), so 32 consecutive ids
// should form a 2-D block that shares as many operand rows as possible in that XCD's 4 MB L2.  With more than BAND tile
// columns a plain "tn fastest" order makes a chunk one M-tile x 32 N-tiles: every chunk streams 32 different W slices (the
// whole weight matrix at N = 10240) through L2 -- measured FETCH_SIZE 3.0 GB against 0.10 GB algorithmic for the level-2
// GEGLU GEMM.  Bands of BAND tile columns make a chunk (32 / BAND) M-tiles x BAND N-tiles.
// Each A row-tile is then fetched once per band instead of once, so banding is only switched on when the weight matrix does
// not fit in L2 anyway (launch3: W > 3 MB; at level 0, W = 1.6 MB, bands measured -7 %, at level 2, 26 MB, +6 %).
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int band, int& tm, int& tn) {
    if (band <= 0 || tiles_n <= band) { tm = id / tiles_n; tn = id - tm * tiles_n; return; }
    const int nb = (tiles_n + band - 1) / band;
    const int per_band = band * tiles_m;
    const int k = min(id / per_band, nb - 1);
    const int r = id - k * per_band;
    const int w = k == nb - 1 ? tiles_n - k * band : band;
    tm = r / w;
    tn = k * band + (r - tm * w);
}

template <int MODE, int EPI>
__global__ __launch_bounds__(64 * NW, 2) void gemm3_kernel(const GemmP p, const SkP sk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // epilogue family: GEGLU (EPI & 8) and the conv modes without lo8 operands go through a wave-private LDS patch; dense GEMMs
    // and everything that carries the split residual stream use the LDS-free direct epilogue (permuted W staging)
    constexpr bool DIRECT = (EPI & 8) == 0 && (MODE == EW_A_DENSE || (EPI & 16));
#ifndef EW_G3_RESLDS
#define EW_G3_RESLDS 1          /* round 5: residual operands of the DIRECT epilogue come in through the LDS (see the RES_LDS epilogue below) */
#endif
    constexpr bool RB_INIT = DIRECT && (EPI & 1);
    // DIRECT variants with residual operands: the epilogue takes both LDS stages as landing zones of its residual tiles (LDS-DMA, 8 KB per
    // wave, operand and row fragment: 60-120 KB in flight per CU instead of the ~24 KB two VGPR operand sets allowed), so the first
    // K-tile of the NEXT work item is not prefetched during the last K-tile of this one but staged inside the epilogue.
    constexpr bool RES_LDS = bool(EW_G3_RESLDS) && DIRECT && (EPI & 6) != 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);

    // ---- tile sequence of this persistent block: step i -> tile id i*G + (b%8)*(G/8) + b/8  (XCD-contiguous chunks)
    const int G = gridDim.x;
    const int total_tiles = p.tiles_m * p.tiles_n;
    const int seq0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int nk = p.K / BK;
    // Work items of this block: [stream-K items] then [data-parallel tiles i*G + seq0].  With the stream-K tail on, the last
    // sk_tiles tiles (ids >= sk_dp_rounds*G) form one stream of sk_tiles*nk K-tile units cut into G equal ranges; range seq0
    // starts with the TAIL of a tile (k0 > 0: this block contributes its partial accumulators to slot seq0 -- first thing it
    // does), continues with whole tiles and ends with the HEAD of a tile (k1 < nk: this block finishes that tile with the
    // partial of block seq0+1 -- which that block wrote at its very start, so the wait is short by construction).
    // (the dense variant with row-bias + two split residuals, <0,23>, is compiled without the tail split: its epilogue is already
    // over the register budget and the extra code doubled its spills: 4.6 -> 5.9 ms per forward)
    constexpr bool SK_OK = !(MODE == EW_A_DENSE && EPI == 23);
    const int sk_tiles = SK_OK ? sk.tiles : 0;
    const int sk_U = sk_tiles * nk;
    const int sk_u0 = sk_tiles ? (int)((long long)seq0 * sk_U / G) : 0;
    const int sk_u1 = sk_tiles ? (int)((long long)(seq0 + 1) * sk_U / G) : 0;
    const int sk_tA = sk_u0 / nk, sk_kA = sk_u0 - sk_tA * nk, sk_tB = sk_u1 / nk, sk_kB = sk_u1 - sk_tB * nk;
    const int n_sk = sk_tiles ? (sk_tB - sk_tA) + (sk_kB > 0 ? 1 : 0) : 0;
    const int sk_id0 = sk.dp_rounds * G;
    const int n_dp = sk_tiles ? sk.dp_rounds : (seq0 < total_tiles ? (total_tiles - 1 - seq0) / G + 1 : 0);
    const int V = (sk_u1 - sk_u0) + n_dp * nk;                         // K-tile stream length of this block
    if (V == 0) return;
    // The items (tile id, first K-tile, end K-tile) go into a small LDS table behind the two stages: the loader and the MFMA
    // stream each read one entry per tile.  (Computing them in place -- a branchy expression inside the loader lambda -- kept
    // hipcc from promoting the lambdas' captured state to registers: 600 bytes of scratch per lane.)
    int* const items = (int*)(smem + 2 * STAGE);
    auto stage_base = [&](int st) __attribute__((always_inline)) -> char* { return smem + st * STAGE; };
    for (int w = threadIdx.x; w < n_sk + n_dp; w += 64 * NW) {
        int id, k0 = 0, k1 = nk;
        if (w < n_sk) {
            const int t = sk_tA + w;
            id = sk_id0 + t;
            if (w == 0) k0 = sk_kA;
            if (t == sk_tB) k1 = sk_kB;
        } else {
            id = (w - n_sk) * G + seq0;
        }
        *(int4*)(items + 4 * w) = make_int4(id, k0, k1, 0);
    }
    __syncthreads();
#define EW3_GET_ITEM(w_, id_, k0_, k1_)                                              \
    do {                                                                             \
        const int4 it__ = *(const int4*)(items + 4 * (w_));                          \
        id_ = __builtin_amdgcn_readfirstlane(it__.x);                                \
        k0_ = __builtin_amdgcn_readfirstlane(it__.y);                                \
        k1_ = __builtin_amdgcn_readfirstlane(it__.z);                                \
    } while (0)

    const int srow = lane >> 3;
    const int slot = (lane & 7) ^ srow;

    // ---------------- loader state (one K-tile ahead of the MFMA stream, across output tiles) ----------------
    constexpr int NTAP = MODE == EW_A_CONV3X3 ? 9 : (MODE == EW_A_CONVT3 ? 3 : 1);
    int ld_w = 0, ld_kt = 0, ld_k1 = 0, ld_tap = 0, ld_cc = 0;   // ld_kt == ld_k1: the next stage_begin opens work item ld_w
    int a_ctr[GA];                         // centre-tap pixel (row) index in the source tensors, relative to the wave's reference pixel a_ref
    int a_mask[GA];                        // bits 0..8 tap validity, bits 16..27 upsample (dy,dx) codes
    int a_ref = 0;                         // wave-uniform reference pixel: every a_ctr[i] (+ an upsample delta) of this wave is >= 0 relative to it
    unsigned b_off = 0;                    // byte offset of this lane's W row (first piece) + 16-byte slot inside the tile column's W slice; piece j adds j*NW*8 rows
    int w_n0 = 0;                          // first W row of the tile column (wave-uniform)

    auto loader_new_tile = [&](const int id) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(id, p.tiles_m, p.tiles_n, p.band, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        // per-lane constants are re-derived from an opaque copy of the lane id: hoisted out of the K-tile stream they would
        // be live across the main loop, where every VGPR is taken, and come back as scratch reloads (each with a vmcnt(0))
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int srow_o = lane_o >> 3, slot_o = (lane_o & 7) ^ srow_o;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            int m = m0 + (wave + NW * i) * 8 + srow_o;
            m = m < p.M ? m : p.M - 1;
            int ctr, mask = 1, dcode = 0;
            if constexpr (MODE == EW_A_CONV3X3) {
                const int hw = p.h_out * p.w_out;
                const int img = m / hw, rem = m - img * hw;
                const int oy = rem / p.w_out, ox = rem - oy * p.w_out;
                const int hlim = p.upsample ? 2 * p.h_in : p.h_in, wlim = p.upsample ? 2 * p.w_in : p.w_in;
                const int cy = oy * p.stride + p.conv_shift, cx = ox * p.stride + p.conv_shift;               // centre tap, in (possibly upsampled) input coords
                mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
                    if (iy >= 0 && iy < hlim && ix >= 0 && ix < wlim) mask |= 1 << t;
                }
                if (p.upsample) {
                    const int sy = cy >> 1, sx = cx >> 1;                        // source pixel of the centre tap
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        dcode |= ((((cy + k - 1) >> 1) - sy) + 1) << (2 * k);        // dy(ky) in {-1,0,1} -> 2 bits
                        dcode |= ((((cx + k - 1) >> 1) - sx) + 1) << (6 + 2 * k);    // dx(kx)
                    }
                    ctr = (img * p.h_in + sy) * p.w_in + sx;
                } else {
                    ctr = (img * p.h_in + cy) * p.w_in + cx;
                }
            } else if constexpr (MODE == EW_A_CONVT3) {
                const int tp = p.tT * p.tP;
                const int bb = m / tp, rem = m - bb * tp;
                const int t = rem / p.tP, x = rem - t * p.tP;
                mask = (t > 0 ? 1 : 0) | 2 | (t + 1 < p.tT ? 4 : 0);
                ctr = (bb * p.tT + t) * p.tP + x;
            } else {
                ctr = m;
            }
            a_mask[i] = mask | (dcode << 16);
            a_ctr[i] = ctr;
        }
        // W row staged at LDS row R = (wave + 8j)*8 + srow.  GEGLU keeps the identity map (host-interleaved value / gate
        // blocks); every other epilogue stages a PERMUTED row so that the accumulators of fragment pair (2q, 2q+1) of one
        // lane are 8 CONSECUTIVE output columns: fragment jj, row i = 4*fks + e  <-  column (jj>>1)*32 + fks*8 + (jj&1)*4 + e.
        // With u = wave + 8j that is column 64*j + f(wave, srow): the same 64-row stride per piece as the identity map.
        int wrow = wave * 8 + srow_o;
        if constexpr (DIRECT)
            wrow = (wave >> 2) * 32 + ((wave >> 1) & 1) * 4 + (2 * (wave & 1) + (srow_o >> 2)) * 8 + (srow_o & 3);
        b_off = (unsigned)((wrow * p.K + slot_o * 8) * 2);                              // N % 320 == 0: every W row exists
        w_n0 = n0;
        {
        // reference pixel of the wave: lane 0 / piece 0 owns the wave's first row, and the source pixel index grows with the row -- except under the
        // nearest-x2 upsample, where two output rows share a source row (the pixel index falls back by < w_in) and the per-row tap deltas reach w_in + 1 more
        a_ref = __builtin_amdgcn_readfirstlane(a_ctr[0]);
        if constexpr (MODE == EW_A_CONV3X3) { if (p.upsample) a_ref -= 2 * p.w_in + 1; }
#pragma unroll
        for (int i = 0; i < GA; ++i) a_ctr[i] -= a_ref;
        }
    };

    // staging of one K-tile = stage_begin (wave-uniform source selection, advances the stream counters) + NP DMA pieces
    const f16* st_base = p.a;
    int st_tap = 0, st_ld = 0, st_ch = 0;
    // Loader addressing (round 6): the pieces are `buffer_load_dwordx4 ... offen lds`.  One A descriptor per K-tile: base = the wave's reference pixel's
    // row at this tap / channel chunk (may lie outside the tensor on border tiles: never dereferenced there), a 2 GB window; a lane's offset is
    // (its pixel - reference) * row bytes + slot -- one v_mad_u32_u24 -- and a padding tap gets an offset outside the window, for which the hardware writes
    // ZEROS to LDS (tools/experiments/buffer_lds/oob_test.hip): no zero page, no 64-bit pointer arithmetic (v_mad_i64_i32 + 2 v_lshl_add_u64 + a 64-bit
    // select per piece before) on the VALU, which shares its issue slots with the MFMAs.  W: descriptor = the tile column's K-tile slice, per-lane offset
    // constant per tile, piece offset in the instruction's SGPR.  VALU instructions per K-tile in the step loop: dense 18 -> 8, 3x3 conv 85 -> 53,
    // temporal conv 49 -> 26; -1.0 ms per forward in a three-round A/B on one box.
    __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t st_wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);     // W: K-tile slice of the tile column's rows
    unsigned st_ld2 = 0;
    char* st_buf = smem;
    auto stage_begin = [&](char* buf) __attribute__((always_inline)) {
        bool opened = false;
        if (ld_kt == ld_k1) {
            opened = true;
            int id, k0;
            EW3_GET_ITEM(ld_w, id, k0, ld_k1);
            ++ld_w;
            loader_new_tile(id);
            ld_kt = k0;
            const int chunk = k0 / NTAP;                                   // K order: channel chunk major, tap minor
            ld_tap = k0 - chunk * NTAP;
            ld_cc = chunk * BK;
        }
        st_buf = buf;
        st_tap = ld_tap;
        const int cc = ld_cc;
        const bool second = cc >= p.c1;
        st_base = second ? p.a2 : p.a;
        st_ld = second ? p.lda2 : p.lda;
        st_ch = second ? cc - p.c1 : cc;
        int dpix = 0;                                                       // wave-uniform tap delta in pixels
        if constexpr (MODE == EW_A_CONV3X3) dpix = (st_tap / 3 - 1) * p.w_in + (st_tap % 3 - 1);
        else if constexpr (MODE == EW_A_CONVT3) dpix = (st_tap - 1) * p.tP;
        if constexpr (MODE == EW_A_CONV3X3) { if (p.upsample) dpix = 0; }          // per-row deltas instead (stage_piece)
        const char* ua = (const char*)st_base + ((long long)(a_ref + dpix) * st_ld + st_ch) * 2;
        st_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ua, 0, 0x7fffffff, 0x00020000);
        st_ld2 = (unsigned)st_ld * 2u;
        st_wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w + ((size_t)w_n0 * p.K + (size_t)ld_kt * BK) * 2), 0, 0x7fffffff, 0x00020000);
        // K order: channel-chunk major, tap minor (the taps of a 64-channel chunk re-hit the same lines in L2 / TCP)
        if (++ld_tap == NTAP) { ld_tap = 0; ld_cc += BK; }
        ++ld_kt;
    };
    auto stage_piece = [&](int k) __attribute__((always_inline)) {     // k is a compile-time constant after unrolling
        if (k < GA) {
            const int i = k;
            unsigned rel = (unsigned)a_ctr[i];
            if constexpr (MODE == EW_A_CONV3X3) {
                if (p.upsample) {                                           // per-row deltas (nearest-x2 source coordinates)
                    const int dc = a_mask[i] >> 16;
                    const int dy = ((dc >> (2 * (st_tap / 3))) & 3) - 1, dx = ((dc >> (6 + 2 * (st_tap % 3))) & 3) - 1;
                    rel = (unsigned)(a_ctr[i] + dy * p.w_in + dx);
                }
            }
            unsigned voff = __umul24(rel, st_ld2) + (unsigned)(slot * 16);          // < 2^24 pixels and row bytes per tile: one full-rate instruction
            if constexpr (MODE != EW_A_DENSE) voff = ((a_mask[i] >> st_tap) & 1) ? voff : 0x80000000u;      // padding tap: outside the window -> zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st_rsrc, (lptr_t)(st_buf + (wave + NW * i) * 1024), 16, (int)voff, 0, 0, 0);
        } else {
            const int j = k - GA;           // piece j = 64 W rows further: the instruction's scalar offset
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st_wrsrc, (lptr_t)(st_buf + A_BYTES + (wave + NW * j) * 1024), 16, (int)b_off, j * (NW * 8 * 2) * p.K, 0, 0);
        }
    };

    // ---------------- fragment geometry ----------------
    const int frow = lane & 15, fks = lane >> 4, sw = frow & 7;
    int a_rd[2], b_rd[2];                  // per-lane byte offset inside a stage of A fragment 0 / W fragment 0, per k-half
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int so = ((kh * 4 + fks) ^ sw) << 4;
        a_rd[kh] = (wm * WM + frow) * 128 + so;
        b_rd[kh] = A_BYTES + (wn * WN + frow) * 128 + so;
    }

    f32x4 acc[FM][FN];
    // Accumulators of a work item start as the bias of its tile column (every row fragment the same 4 columns per lane and
    // fragment), or as zeros for the TAIL of a stream-K tile (k0 > 0: another block owns the head, and the bias with it).
    auto init_acc = [&](const int id, const int k0) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(id, p.tiles_m, p.tiles_n, p.band, tm, tn);
        int lane_o = tid & 63;                       // opaque copy: keeps the bias addresses out of the main loop's live set
        asm volatile("" : "+v"(lane_o));
        const int fks_o = lane_o >> 4;
        const f16* bsrc = (p.bias && k0 == 0) ? p.bias + tn * BN + wn * WN + (DIRECT ? fks_o * 8 : fks_o * 4) : p.zero_page;
        const int on = (p.bias && k0 == 0) ? 1 : 0;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            // DIRECT (permuted W staging): fragment j, element e <-> column (j >> 1) * 32 + fks * 8 + (j & 1) * 4 + e; otherwise j * 16 + fks * 4 + e
            const int col = DIRECT ? (j >> 1) * 32 + (j & 1) * 4 : j * 16;
            const f16x4 b4 = *(const f16x4*)(bsrc + col * on);
            const f32x4 b = {(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[i][j] = b;
        }
        if constexpr (RB_INIT) {
            // + row-bias of the row group of each of the lane's FM rows (same 4 / 8 consecutive columns as the bias above)
            const int frow_o = lane_o & 15;
            const int onr = (p.rowbias && k0 == 0) ? 1 : 0;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = min(tm * BM + wm * WM + i * 16 + frow_o, p.M - 1);
                const int g = row / p.rows_per_group;
                const f16* rsrc = onr ? p.rowbias + (size_t)g * p.ld_rowbias + tn * BN + wn * WN + fks_o * 8 : p.zero_page;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int col = (j >> 1) * 32 + (j & 1) * 4;
                    const f16x4 r4 = *(const f16x4*)(rsrc + col * onr);
                    acc[i][j] += (f32x4){(float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]};
                }
            }
        }
    };
    f16x8 af[2][FM];                       // A fragments of the two k-halves
    f16x8 bfr[4];                          // W fragment ring: step t consumes bfr[t & 3]

    // ---------------- prologue ----------------
    stage_begin(stage_base(0));
#pragma unroll
    for (int k = 0; k < NP; ++k) stage_piece(k);
    int staged = 1;
    EW3_WAIT_VM0();
    EW3_FENCE();
    __builtin_amdgcn_s_barrier();
    EW3_FENCE();
    int cur_w = 0, cur_id, cur_kt, cur_k1;
    EW3_GET_ITEM(0, cur_id, cur_kt, cur_k1);
    cur_w = 1;
    {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[0][i] = EW3_LDS(smem + a_rd[0] + i * 2048);
    }
#pragma unroll
    for (int j = 0; j < PD; ++j) bfr[j] = EW3_LDS(stage_base(0) + b_rd[0] + j * 2048);

    init_acc(cur_id, cur_kt);
    bool cur_tail = cur_kt > 0;                      // stream-K: this item is the tail of a tile another block finishes
    int s_cur = 0;                                   // ring slot of stream position v
    for (int v = 0; v < V; ++v) {
        const char* cur = stage_base(s_cur);
        char* nxt = stage_base(s_cur ^ 1);
        const bool tile_end = cur_kt == cur_k1 - 1;
        const bool pend = staged < V;                // K-tile v+1 exists: stage it into the other slot during steps 0..8
        // RES_LDS: at a tile end both stages belong to the epilogue's residual tiles; K-tile v+1 (the first of the next work item)
        // is staged inside the epilogue instead (tile-end section below)
        const bool pend_now = pend && !(RES_LDS && tile_end);
        if (pend_now) { stage_begin(nxt); ++staged; }
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            const int kh = t / FN, j = t - kh * FN;
            // ---- reads: W fragment of step t+2 (from the next K-tile once past the barrier), A fragments of the next k-half
            if (t + PD < NSTEP) {
                const int kh2 = (t + PD) / FN, j2 = (t + PD) - kh2 * FN;
                bfr[(t + PD) & 3] = EW3_LDS(cur + b_rd[kh2] + j2 * 2048);
            }
            if (t < FM) af[1][t] = EW3_LDS(cur + a_rd[1] + t * 2048);
            if (t >= NSTEP - PD) {
                // first fragments of the next K-tile -- except at a tile end: the epilogue needs the registers (160 live
                // accumulators), so they are read after it instead
                if (!tile_end) {
                    bfr[(t + PD) & 3] = EW3_LDS(nxt + b_rd[0] + (t + PD - NSTEP) * 2048);
                    if (t >= NSTEP - 2) {
                        const int i0 = (t - (NSTEP - 2)) * 2;
                        af[0][i0] = EW3_LDS(nxt + a_rd[0] + i0 * 2048);
                        af[0][i0 + 1] = EW3_LDS(nxt + a_rd[0] + (i0 + 1) * 2048);
                    }
                }
            }
            if (t < NP) {
                if (pend_now) stage_piece(t);
            }
            EW3_PIN();
#pragma unroll
            for (int i = 0; i < FM; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[t & 3], af[kh][i], acc[i][j], 0, 0, 0);   // D[n][m]
            EW3_PIN();
            if (t == BAR_STEP) {
                // every fragment read of K-tile v has been issued; publish K-tile v+1 and free this slot.  Unconditional
                // (also on the last position, where the prefetched fragments are stale and never used).
                EW3_WAIT_VM0();
                EW3_WAIT_LGKM0();
                EW3_FENCE();
                __builtin_amdgcn_s_barrier();
                EW3_FENCE();
            }
        }
        s_cur ^= 1;
        if (++cur_kt == cur_k1) {
            // ------------------------- end of work item: epilogue of output tile cur_id (or stream-K hand-over) -------------------------
            const int id = cur_id;
            const bool sk_contribute = SK_OK && cur_tail;             // tail of a tile: hand the partial accumulators over, no epilogue
            const bool sk_finish = SK_OK && cur_k1 < nk;              // head of a tile: add the other block's partial, then the epilogue
            if (cur_w < n_sk + n_dp) {
                EW3_GET_ITEM(cur_w, cur_id, cur_kt, cur_k1);
                cur_tail = cur_kt > 0;                       // never true past item 0; kept general
            }
            ++cur_w;
            int tm, tn;
            tile_coords(id, p.tiles_m, p.tiles_n, p.band, tm, tn);
            const bool full = (tm * BM + wm * WM + WM <= p.M);          // N is always full (N % 320 == 0)
            // wave-private fp32 patch in the slot just consumed (free since the barrier of step 17; the DMA of the next
            // K-tile into it is issued by the NEXT position, after the closing barrier below)
            constexpr int CP = WN / 2;                             // columns per pass (80 at BN = 320)
            constexpr int LDP = CP + 4;                            // patch row stride (floats)
            float* patch = (float*)(smem + (s_cur ^ 1) * STAGE) + wave * (16 * LDP);
            const int m_w0 = tm * BM + wm * WM, n_w0 = tn * BN + wn * WN;
            auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;
                int lane = tid & 63;                        // opaque copy: keeps the epilogue's per-lane constants out of the main loop
                asm volatile("" : "+v"(lane));
                const int frow = lane & 15, fks = lane >> 4;
                const f16* bp = p.bias ? p.bias : p.zero_page;
                const f16* rbp = p.rowbias ? p.rowbias : p.zero_page;
                const f16* r1p = p.r1 ? p.r1 : p.zero_page;
                const f16* r2p = p.r2 ? p.r2 : p.zero_page;
                const int mbias = p.bias ? 1 : 0, mrb = p.rowbias ? 1 : 0, m1 = p.r1 ? 1 : 0, m2 = p.r2 ? 1 : 0;
                const int ldrb = p.rowbias ? p.ld_rowbias : 0, ld1 = p.r1 ? p.ld_r1 : 0, ld2 = p.r2 ? p.ld_r2 : 0;
                constexpr int VPR = CP / 8;                 // 16-byte output vectors per row and pass
                constexpr int ITERS = (16 * VPR + 63) / 64;  // 3 (the last one partial: 160 = 2*64 + 32)
                auto is_live = [&](int it) { return it * 64 + lane < 16 * VPR; };
                if constexpr (!DIRECT && (EPI & ~1) == 0) {
                    // No residual operands: bias / row-bias / SiLU / scale are applied in the ACCUMULATOR layout (4 consecutive
                    // columns per lane: 8-byte operand loads of L2-resident vectors) and the result goes through the patch as
                    // fp16 -- half the LDS bytes of the fp32 patch, no conversion after the read-back.  Same values, rounded
                    // once, as the general path below.
                    constexpr bool RB = EPI & 1;
                    constexpr int NH = WN / CP;
                    constexpr int LDH = CP + 8;              // patch row stride in halfs (176 B)
                    f16* patch16 = (f16*)patch;
                    const int rpg = p.rows_per_group;
                    // SiLU / GELU are compiled into the plain dense variants only (the U-Net's SiLU GEMMs have M = 2 ... 25 and run on generation 2,
                    // GELU is CLIP's fc1): in every other variant the dead activation block was ~100 instructions and 3 branches per epilogue step
                    constexpr bool ACT_OK = MODE == EW_A_DENSE && (EPI & ~1) == 0;
                    const bool silu = ACT_OK && p.act == EW_ACT_SILU, gelu = ACT_OK && EPI == 0 && p.act == EW_ACT_GELU;
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int m_l = min(m_w0 + i * 16 + frow, p.M - 1);
                        const int g = RB ? m_l / rpg : 0;
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
#pragma unroll
                            for (int jj = 0; jj < CP / 16; ++jj) {
                                const int j = h * (CP / 16) + jj;
                                const int n = n_w0 + j * 16 + fks * 4;
                                f32x4 x = acc[i][j];
                                if constexpr (RB) {
                                    const f16x4 r4 = *(const f16x4*)((const char*)rbp + (unsigned)(g * ldrb + n * mrb) * 2u);
                                    x += (f32x4){(float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]};
                                }
                                if (silu) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) x[e] = ew_silu(x[e]);
                                } else if (gelu) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) x[e] = ew_gelu(x[e]);
                                }
                                x *= p.c_acc;
                                const f16x4 o4 = {(f16)x[0], (f16)x[1], (f16)x[2], (f16)x[3]};
                                *(f16x4*)(patch16 + frow * LDH + jj * 16 + fks * 4) = o4;
                            }
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int it = 0; it < ITERS; ++it) {
                                const int idx = it * 64 + lane;
                                const int row = is_live(it) ? idx / VPR : 0, c8 = is_live(it) ? (idx - row * VPR) * 8 : 0;
                                const int m = m_w0 + i * 16 + row, n = n_w0 + h * CP + c8;
                                const f16x8 o = *(const f16x8*)(patch16 + row * LDH + c8);
                                if (is_live(it) && (FULL || m < p.M)) *(f16x8*)((char*)p.out + (unsigned)(m * p.ld_out + n) * 2u) = o;
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                } else if constexpr (!DIRECT && (EPI & 8) == 0) {
                    // general LDS-patch epilogue (conv modes without lo8 operands: the direct epilogue below costs them 11-18 spilled
                    // VGPRs, some reloaded inside the K loop -- measured 4-10 % slower than this path)
                    constexpr bool RB = EPI & 1, R1 = EPI & 2, R2 = EPI & 4;
                    constexpr int NH = WN / CP;              // 2 column passes
                    int rowv[ITERS], c8v[ITERS];
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int idx = it * 64 + lane;
                        rowv[it] = is_live(it) ? idx / VPR : 0;
                        c8v[it] = is_live(it) ? (idx - rowv[it] * VPR) * 8 : 0;
                    }
                    // vmcnt is in-order: a load issued after a store cannot be waited for without draining that store, so the
                    // operands of store step s+1 are requested BEFORE the store of step s -- but AFTER step s has consumed
                    // its own operands, into the same registers (one operand set: the 160 live accumulators leave no room
                    // for two).
                    f16x8 bvv, rbv, q1v, q2v;
                    // row-bias group of a row: one boundary at most inside the wave's 64 rows when rows_per_group >= 64
                    const int rpg = p.rows_per_group;
                    const int g0 = min(m_w0, p.M - 1) / rpg;             // wave tiles past the last row must not index a group beyond the last
                    const int gbound = rpg >= WM ? (g0 + 1) * rpg : 0x7fffffff;
                    auto fetch = [&](int i, int h, int it) {
                        const int m = m_w0 + i * 16 + rowv[it], n = n_w0 + h * CP + c8v[it];
                        const int mc = FULL ? m : min(m, p.M - 1);
                        // uniform base + 32-bit byte offset (every operand of this path is < 4 GB): one VGPR per address
                        if constexpr (RB) {
                            const int g = rpg >= WM ? g0 + (mc >= gbound ? 1 : 0) : mc / rpg;
                            rbv = *(const f16x8*)((const char*)rbp + (unsigned)(g * ldrb + n * mrb) * 2u);
                        }
                        if constexpr (R1) q1v = *(const f16x8*)((const char*)r1p + (unsigned)(mc * ld1 + n * m1) * 2u);
                        if constexpr (R2) q2v = *(const f16x8*)((const char*)r2p + (unsigned)(mc * ld2 + n * m2) * 2u);
                    };
                    fetch(0, 0, 0);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
#pragma unroll
                            for (int jj = 0; jj < CP / 16; ++jj) {
                                *(f32x4*)(patch + frow * LDP + jj * 16 + fks * 4) = acc[i][h * (CP / 16) + jj];
                            }
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int it = 0; it < ITERS; ++it) {
                                const int row = rowv[it], c8 = c8v[it];
                                const int m = m_w0 + i * 16 + row, n = n_w0 + h * CP + c8;
                                const f32x4 lo = *(const f32x4*)(patch + row * LDP + c8);
                                const f32x4 hi = *(const f32x4*)(patch + row * LDP + c8 + 4);
                                float vv[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                f16x8 o;
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    if constexpr (RB) vv[e] += (float)rbv[e];
                                }
                                // (no activation in this path: SiLU / GELU problems with residual operands or conv modes run on generation 2)
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    float x = vv[e] * p.c_acc;
                                    if constexpr (R1) x += p.c_r1 * (float)q1v[e];
                                    if constexpr (R2) x += p.c_r2 * (float)q2v[e];
                                    o[e] = (f16)x;
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                if (it + 1 < ITERS) fetch(i, h, it + 1);
                                else if (h + 1 < NH) fetch(i, h + 1, 0);
                                else if (i + 1 < FM) fetch(i + 1, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if (is_live(it) && (FULL || m < p.M)) *(f16x8*)((char*)p.out + (unsigned)(m * p.ld_out + n) * 2u) = o;
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                } else if constexpr ((EPI & 8) == 0) {
                    // Direct epilogue, no LDS: thanks to the permuted W staging (loader_new_tile) the two accumulator
                    // fragments (2q, 2q+1) of a lane are 8 consecutive output columns of row frow -> every operand access
                    // (bias, row-bias, residuals and their lo halves, output hi / lo) is ONE 16-byte access per lane, four
                    // lanes cover a 64-byte row segment, a wave instruction covers 16 rows x 64 B.
                    constexpr bool RB = (EPI & 1) && !RB_INIT, R1 = EPI & 2, R2 = EPI & 4, LO = EPI & 16;      // (RB_INIT: the row-bias is already in the accumulators)
                    const int8_t* r1lp = p.r1_lo ? p.r1_lo : (const int8_t*)p.zero_page;      // lo8 companions (common.h): 1 byte / element
                    const int8_t* r2lp = p.r2_lo ? p.r2_lo : (const int8_t*)p.zero_page;
                    const int m1l = p.r1_lo ? 1 : 0, m2l = p.r2_lo ? 1 : 0;
                    const int ld1l = p.r1_lo ? p.ld_r1 : 0, ld2l = p.r2_lo ? p.ld_r2 : 0;
                    constexpr int NQ = FN / 2;                  // 5 fragment pairs = 5 x 32 columns per wave tile
                    // vmcnt is in-order: a load issued after a store cannot be waited for without draining that store, so the
                    // operands of step s+1 are requested BEFORE the store of step s -- but after step s has consumed its own.
                    constexpr int ED = (EPI == 23) ? 1 : ((R1 || R2 || RB) ? 2 : 1);
                    f16x8 rbv[ED], q1v[ED], q2v[ED];
                    u32x2 q1l[ED], q2l[ED];
                    const int rpg = p.rows_per_group;
                    const int g0 = min(m_w0, p.M - 1) / rpg;             // wave tiles past the last row must not index a group beyond the last
                    const int gbound = rpg >= WM ? (g0 + 1) * rpg : 0x7fffffff;
                    const int ncol0 = n_w0 + fks * 8;
                    auto fetch = [&](const int k) __attribute__((always_inline)) {       // step k = i * NQ + q -> operand set k % ED
                        const int i = k / NQ, q = k - i * NQ, st = k % ED;
                        const int m = m_w0 + i * 16 + frow, n = ncol0 + q * 32;
                        const int mc = FULL ? m : min(m, p.M - 1);
                        // uniform base + 32-bit byte offset (every operand of this path is < 4 GB): one VGPR per address
                        if constexpr (RB) {
                            const int g = rpg >= WM ? g0 + (mc >= gbound ? 1 : 0) : mc / rpg;
                            rbv[st] = *(const f16x8*)((const char*)rbp + (unsigned)(g * ldrb + n * mrb) * 2u);
                        }
                        if constexpr (R1) q1v[st] = *(const f16x8*)((const char*)r1p + (unsigned)(mc * ld1 + n * m1) * 2u);
                        if constexpr (R2) q2v[st] = *(const f16x8*)((const char*)r2p + (unsigned)(mc * ld2 + n * m2) * 2u);
                        if constexpr (R1 && LO) q1l[st] = *(const u32x2*)((const char*)r1lp + (unsigned)(mc * ld1l + n * m1l));
                        if constexpr (R2 && LO) q2l[st] = *(const u32x2*)((const char*)r2lp + (unsigned)(mc * ld2l + n * m2l));
                    };
                    // SiLU / GELU are compiled into the plain dense variants only (the U-Net's SiLU GEMMs have M = 2 ... 25 and run on generation 2,
                    // GELU is CLIP's fc1): in every other variant the dead activation block was ~100 instructions and 3 branches per epilogue step
                    constexpr bool ACT_OK = MODE == EW_A_DENSE && (EPI & ~1) == 0;
                    const bool silu = ACT_OK && p.act == EW_ACT_SILU, gelu = ACT_OK && EPI == 0 && p.act == EW_ACT_GELU;
#pragma unroll
                    for (int k = 0; k < ED; ++k) fetch(k);
#pragma unroll
                    for (int k = 0; k < FM * NQ; ++k) {
                        {
                            const int i = k / NQ, q = k - i * NQ, st = k % ED;
                            const int m = m_w0 + i * 16 + frow, n = ncol0 + q * 32;
                            const f32x4 a0 = acc[i][2 * q], a1 = acc[i][2 * q + 1];
                            float vv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                            f16x8 o;
                            u32x2 ol;
                            int s8[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                if constexpr (RB) vv[e] += (float)rbv[st][e];
                            }
                            if (silu) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) vv[e] = ew_silu(vv[e]);
                            } else if (gelu) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) vv[e] = ew_gelu(vv[e]);
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float x = vv[e] * p.c_acc;
                                if constexpr (R1 && LO) x += p.c_r1 * ew_split_dec(q1v[st][e], ew_sbyte(q1l[st][e >> 2], e & 3));
                                else if constexpr (R1) x += p.c_r1 * (float)q1v[st][e];
                                if constexpr (R2 && LO) x += p.c_r2 * ew_split_dec(q2v[st][e], ew_sbyte(q2l[st][e >> 2], e & 3));
                                else if constexpr (R2) x += p.c_r2 * (float)q2v[st][e];
                                o[e] = (f16)x;
                                if constexpr (LO) s8[e] = ew_split_enc(x, o[e]);
                            }
                            if constexpr (LO) { ol[0] = ew_pack4(s8[0], s8[1], s8[2], s8[3]); ol[1] = ew_pack4(s8[4], s8[5], s8[6], s8[7]); }
                            __builtin_amdgcn_sched_barrier(0);
                            if (k + ED < FM * NQ) fetch(k + ED);
                            __builtin_amdgcn_sched_barrier(0);
                            if ((FULL || m < p.M)) {
                                *(f16x8*)((char*)p.out + (unsigned)(m * p.ld_out + n) * 2u) = o;
                                if constexpr (LO) {
                                    if (p.out_lo) *(u32x2*)((char*)p.out_lo + (unsigned)(m * p.ld_out + n)) = ol;
                                }
                            }
                        }
                    }
                } else {
                    // GEGLU: staged column blocks of 32 = [16 value | 16 gate] -> fragment 2q holds the values, 2q+1 the gates of
                    // the SAME (row, column) positions in the SAME lane: value*gelu(gate) in registers, WN/2 = 80 output columns
                    f32x4 bq[FN];
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        bq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
#pragma unroll
                        for (int q = 0; q < FN / 2; ++q) {
                            const f32x4 va = acc[i][2 * q] + bq[2 * q], gg = acc[i][2 * q + 1] + bq[2 * q + 1];
                            const f32x2 o01 = ew_vgelu2((f32x2){va[0], va[1]}, (f32x2){gg[0], gg[1]});
                            const f32x2 o23 = ew_vgelu2((f32x2){va[2], va[3]}, (f32x2){gg[2], gg[3]});
                            const f16x4 o4 = {(f16)o01[0], (f16)o01[1], (f16)o23[0], (f16)o23[1]};
                            *(f16x4*)((f16*)patch + frow * (CP + 8) + q * 16 + fks * 4) = o4;      // fp16 patch, row stride 176 B
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int it = 0; it < ITERS; ++it) {
                            const int idx = it * 64 + lane;
                            const int row = is_live(it) ? idx / VPR : 0, c8 = is_live(it) ? (idx - row * VPR) * 8 : 0;
                            const int m = m_w0 + i * 16 + row;
                            const f16x8 o = *(const f16x8*)((const f16*)patch + row * (CP + 8) + c8);
                            const int no = (n_w0 >> 1) + c8;
                            // non-temporal: the 4C-wide GEGLU output (1.2 GB at level 0) only evicts the operands from L2
                            // (+5..6 % measured here; the same hint on the other epilogues measured -1..-18 %)
                            if (is_live(it) && (FULL || m < p.M)) __builtin_nontemporal_store(o, (f16x8*)(p.out + (size_t)m * p.ld_out + no));
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            };
            bool res_run = false;                    // RES_LDS: run the epilogue below (not a stream-K contributor)
            if (sk_finish) {
                // stream-K finisher: block seq0+1 wrote its partial of this tile as the first thing it did
                if (tid == 0) {
                    unsigned* fl = sk.flags + seq0 + 1;
                    int spins = 0;
                    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1 << 24)) {             // ~10 s: never hang the device; poison the flag word so the host sees it
                            // (which launch: its epoch goes next to the poison word; the host keeps a ring of launch descriptions per workspace)
                            __hip_atomic_store(sk.flags + 1025, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(sk.flags + 1024, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    // consumed: clear the flag, so that a replay of this launch with the SAME epoch (a captured hipGraph) waits for
                    // the partial of the replay and not for the stale flag of the previous run
                    __hip_atomic_store(fl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                EW3_FENCE();
                __builtin_amdgcn_s_barrier();
                EW3_FENCE();
                // The partial comes in through the LDS-DMA path into the stage the K loop has just released (9 KB per wave: eight
                // 1 KB pieces in flight, no VGPRs), then one fragment at a time is read back and added: five round trips to memory,
                // and no register demand on top of the 160 live accumulators (ten fragments in VGPRs cost <0,23> 18 more spills).
                int lane_o = tid & 63;
                asm volatile("" : "+v"(lane_o));
                const f32x4* wsp = (const f32x4*)sk.ws + (size_t)(seq0 + 1) * (FM * FN * 64 * NW) + wave * 64 + lane_o;
                constexpr int SKU = 8;
                char* const stg = stage_base(s_cur ^ 1) + wave * (SKU * 1024);
#pragma unroll
                for (int b0 = 0; b0 < FM * FN; b0 += SKU) {
#pragma unroll
                    for (int u = 0; u < SKU; ++u) glds16((const f16*)(wsp + (b0 + u) * (64 * NW)), stg + u * 1024);
                    EW3_WAIT_VM0();
                    EW3_FENCE();
#pragma unroll
                    for (int u = 0; u < SKU; ++u) {
                        const int f = b0 + u;
                        acc[f / FN][f % FN] += *(const f32x4*)(stg + u * 1024 + lane_o * 16);
                    }
                    EW3_WAIT_LGKM0();
                    EW3_FENCE();
                }
                // the staging regions (8 KB per wave) overlap the other waves' epilogue patches (5.25 KB per wave) in the same stage
                __builtin_amdgcn_s_barrier();
                EW3_FENCE();
            }
            if (sk_contribute) {
                // stream-K contributor: accumulators -> slot seq0 of the uncached workspace in [fragment][thread] order (16 B per
                // lane, 1 KB per wave instruction), then publish.  Every thread drains its own stores (vmcnt counts stores on
                // gfx9) before the barrier; one thread raises the flag after it.
                int tid_o = tid;                              // opaque copy: keeps the workspace address out of the main loop's live set
                asm volatile("" : "+v"(tid_o));
                f32x4* wsp = (f32x4*)sk.ws + (size_t)seq0 * (FM * FN * 64 * NW) + tid_o;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        wsp[(i * FN + j) * (64 * NW)] = acc[i][j];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                EW3_WAIT_VM0();
                EW3_FENCE();
                __builtin_amdgcn_s_barrier();
                EW3_FENCE();
                if (tid == 0) __hip_atomic_store(sk.flags + seq0, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if constexpr (!RES_LDS) {
                if (full) epilogue(std::true_type{}); else epilogue(std::false_type{});
                // the patch lives in the slot the next position's DMA will overwrite
                EW3_WAIT_LGKM0();
                EW3_FENCE();
                __builtin_amdgcn_s_barrier();
                EW3_FENCE();
            } else {
                res_run = true;
            }
            if constexpr (RES_LDS) {
                // ---- DIRECT epilogue with the residual tiles landing in LDS (round 5) ----
                // Operand tile of one row fragment i (16 rows x 160 columns of this wave): hi plane 5 KB = 5 LDS-DMA instructions
                // (instruction q, lane L: row L & 15, the 16-byte chunk q * 4 + (L >> 4) -- exactly the lane's operand of epilogue
                // step (i, q), so the read-back is one conflict-free ds_read_b128 at lane * 16), lo8 plane 2.5 KB = 3 instructions
                // ([10 chunks of 16 columns][16 rows] x 16 B, the last one on lanes 0-31 only; read back as ds_read_b64).
                // Landing zones: one per wave (STAGE / 8 = 9 KB) in each of the two stages (wave-private: no workgroup barrier between
                // the DMA and the read-back, only the wave's own counted vmcnt).  One residual: fragments i alternate between the
                // two stages, two fragments in flight (B0 B1 | S0 | B2 | S1 | B3 | S2 | KT | S3); two residuals: r1 in one stage, r2
                // in the other, one fragment in flight.  vmcnt retires in order: a wait for batch Bi names the instructions issued
                // AFTER it (the stores Sj of 10 instructions per fragment, later batches of 8 / 16, the 9 pieces KT of the next
                // K-tile) -- exact on full tiles; on edge tiles (stores masked, possibly skipped) only the DMA instructions count.
                constexpr bool R1 = EPI & 2, R2 = EPI & 4, LO = EPI & 16;
                constexpr int NQ = FN / 2;
                constexpr int NOPD = (R1 && R2) ? 2 : 1;                 // residual operands
                constexpr int HI_B = NQ * 1024;                           // hi plane of one fragment tile (5 KB at BN = 320, 4 KB at 256)
                constexpr int LO_CH = WN / 16, LO_I = (LO_CH + 3) / 4;    // lo8 plane: 16-column chunks per row, LDS-DMA instructions (the last one partial at BN = 320)
                constexpr int ZONE = STAGE / NW;                          // landing zone of one wave in one stage
                static_assert(HI_B + LO_I * 1024 <= ZONE, "landing zone");
                constexpr int BI = NOPD * (NQ + (LO ? LO_I : 0));         // DMA instructions per batch
                constexpr int SI = NQ * (LO ? 2 : 1);                     // store instructions per row fragment
                char* const stage_next = smem + s_cur * STAGE;            // stage of stream position v+1 (free since the barrier of position v-1)
                char* const stage_done = smem + (s_cur ^ 1) * STAGE;      // stage just consumed (free since this position's barrier)
                const bool deferred = pend;                               // K-tile v+1 exists and has not been staged yet
                const bool exact = full && (!LO || p.out_lo);       // every store instruction of a fragment is issued
#define EW3_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
                auto res_phase = [&](auto phase_tag) __attribute__((always_inline)) {
                    constexpr int PH = decltype(phase_tag)::value;        // 0: fragments 0 .. FM-2 (one residual) or all (two), 1: fragment FM-1 (one residual)
                    int lane_e = tid & 63;                                // opaque copy: keeps the per-lane constants out of the main loop
                    asm volatile("" : "+v"(lane_e));
                    const int frow = lane_e & 15, fks = lane_e >> 4;
                    const unsigned hi_rd = lane_e * 16;
                    const unsigned lo_rd = HI_B + (((fks >> 1) * 16 + frow) * 16) + (fks & 1) * 8;
                    const char* r1p = (const char*)(p.r1 ? p.r1 : p.zero_page);
                    const char* r2p = (const char*)(p.r2 ? p.r2 : p.zero_page);
                    const char* r1lp = (const char*)(p.r1_lo ? (const void*)p.r1_lo : (const void*)p.zero_page);
                    const char* r2lp = (const char*)(p.r2_lo ? (const void*)p.r2_lo : (const void*)p.zero_page);
                    const int m1 = p.r1 ? 1 : 0, m2 = p.r2 ? 1 : 0, m1l = p.r1_lo ? 1 : 0, m2l = p.r2_lo ? 1 : 0;
                    const int ld1 = p.r1 ? p.ld_r1 : 0, ld2 = p.r2 ? p.ld_r2 : 0, ld1l = p.r1_lo ? p.ld_r1 : 0, ld2l = p.r2_lo ? p.ld_r2 : 0;
                    char* const za = stage_next + wave * ZONE;            // landing zones of this wave
                    char* const zb = stage_done + wave * ZONE;
                    // one operand's tile of row fragment i -> zone
                    auto issue1 = [&](const char* rp, const char* rlp, int ld, int mm, int ldl, int mml, int i, char* zone) __attribute__((always_inline)) {
                        const int mc = min(m_w0 + i * 16 + frow, p.M - 1);
                        const char* src = rp + (unsigned)(mc * ld + (n_w0 + fks * 8) * mm) * 2u;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) glds16((const f16*)(src + q * 64 * mm), zone + q * 1024);
                        if constexpr (LO) {
                            const char* srcl = rlp + (unsigned)(mc * ldl + (n_w0 + fks * 16) * mml);
#pragma unroll
                            for (int tt = 0; tt < LO_I; ++tt) {
                                if (tt * 4 + 4 <= LO_CH) glds16((const f16*)(srcl + tt * 64 * mml), zone + HI_B + tt * 1024);
                                else if (lane_e < (LO_CH - tt * 4) * 16) glds16((const f16*)(srcl + tt * 64 * mml), zone + HI_B + tt * 1024);
                            }
                        }
                    };
                    auto issue = [&](int i) __attribute__((always_inline)) {
                        if constexpr (NOPD == 2) {
                            issue1(r1p, r1lp, ld1, m1, ld1l, m1l, i, za);
                            issue1(r2p, r2lp, ld2, m2, ld2l, m2l, i, zb);
                        } else if constexpr (R1) {
                            issue1(r1p, r1lp, ld1, m1, ld1l, m1l, i, (i & 1) ? zb : za);
                        } else {
                            issue1(r2p, r2lp, ld2, m2, ld2l, m2l, i, (i & 1) ? zb : za);
                        }
                    };
                    auto process = [&](int i) __attribute__((always_inline)) {
                        const char* z1 = NOPD == 2 ? za : ((i & 1) ? zb : za);     // first (or only) operand
                        const char* z2 = zb;                                        // second operand (NOPD == 2)
                        f16x8 h1[NQ], h2[NQ];
                        u32x2 l1[NQ], l2[NQ];
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            h1[q] = *(const f16x8*)(z1 + hi_rd + q * 1024);
                            if constexpr (LO) l1[q] = *(const u32x2*)(z1 + lo_rd + q * 512);
                            if constexpr (NOPD == 2) {
                                h2[q] = *(const f16x8*)(z2 + hi_rd + q * 1024);
                                if constexpr (LO) l2[q] = *(const u32x2*)(z2 + lo_rd + q * 512);
                            }
                        }
                        const int m = m_w0 + i * 16 + frow;
                        const bool live = (m < p.M);
                        const unsigned ob = (unsigned)(m * p.ld_out + n_w0 + fks * 8);
                        const float ca = p.c_acc, cA = (NOPD == 2 || R1) ? p.c_r1 : p.c_r2, cB = p.c_r2;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const f32x4 a0 = acc[i][2 * q], a1 = acc[i][2 * q + 1];
                            const float vv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                            f16x8 o;
                            u32x2 ol;
                            int s8[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float x = vv[e] * ca;
                                if constexpr (LO) x += cA * ew_split_dec(h1[q][e], ew_sbyte(l1[q][e >> 2], e & 3));
                                else x += cA * (float)h1[q][e];
                                if constexpr (NOPD == 2) {
                                    if constexpr (LO) x += cB * ew_split_dec(h2[q][e], ew_sbyte(l2[q][e >> 2], e & 3));
                                    else x += cB * (float)h2[q][e];
                                }
                                o[e] = (f16)x;
                                if constexpr (LO) s8[e] = ew_split_enc(x, o[e]);
                            }
                            if constexpr (LO) { ol[0] = ew_pack4(s8[0], s8[1], s8[2], s8[3]); ol[1] = ew_pack4(s8[4], s8[5], s8[6], s8[7]); }
                            if (live) {
                                *(f16x8*)((char*)p.out + (ob + q * 32) * 2u) = o;
                                if constexpr (LO) {
                                    if (p.out_lo) *(u32x2*)((char*)p.out_lo + (ob + q * 32)) = ol;
                                }
                            }
                        }
                    };
                    if constexpr (PH == 0) {
                        if constexpr (NOPD == 1) {
                            issue(0);
                            issue(1);
#pragma unroll
                            for (int i = 0; i < FM - 1; ++i) {
                                // batch i landed?  issued after it so far: B(i+1), and the stores of fragment i-1 (between B(i) .. no: see order)
                                // order: B0 B1 | S0 B2 | S1 B3 | S2 ...  -> newer than B(i): i == 0: B1;  i >= 1: S(i-1), B(i+1)
                                if (i == 0 || !exact) EW3_VMCNT(BI); else EW3_VMCNT(BI + SI);
                                process(i);
                                EW3_FENCE();
                                if (i + 2 < FM) issue(i + 2);
                            }
                        } else {
                            // two residuals: both stages hold operands of EVERY fragment, so all FM fragments are done here and K-tile
                            // v+1 is staged after the last one
                            issue(0);
#pragma unroll
                            for (int i = 0; i < FM; ++i) {
                                EW3_VMCNT(0);
                                process(i);
                                EW3_FENCE();
                                if (i + 1 < FM) issue(i + 1);
                            }
                        }
                    } else if constexpr (NOPD == 1) {
                        // fragment FM-1 (its zone is in stage_done): newer than its batch are the stores of fragment FM-2 and, when staged,
                        // the 9 pieces of K-tile v+1
                        if (exact) { if (deferred) EW3_VMCNT(SI + NP); else EW3_VMCNT(SI); }
                        else { if (deferred) EW3_VMCNT(NP); else EW3_VMCNT(0); }
                        process(FM - 1);
                    }
                };
                if (res_run) res_phase(std::integral_constant<int, 0>{});
                if (deferred) {
                    // every wave has read what it needs from stage_next (fragments 0 .. FM-2 done): stage K-tile v+1 there
                    EW3_WAIT_LGKM0();
                    EW3_FENCE();
                    __builtin_amdgcn_s_barrier();
                    EW3_FENCE();
                    stage_begin(stage_next);
                    ++staged;
#pragma unroll
                    for (int k = 0; k < NP; ++k) stage_piece(k);
                }
                if (res_run) res_phase(std::integral_constant<int, 1>{});
                // K-tile v+1 landed (newer: the stores of the last fragment), every wave done with its landing zones
                if (NOPD == 1 && res_run && exact) EW3_VMCNT(SI); else EW3_WAIT_VM0();
                EW3_WAIT_LGKM0();
                EW3_FENCE();
                __builtin_amdgcn_s_barrier();
                EW3_FENCE();
            }
            // Every path above only READS the accumulators; they are cleared here, once, for the next work item (clearing them
            // inside each path gave the allocator a three-way merge of 160 registers: copies and ~280 spilled VGPRs).
            init_acc(cur_id, cur_kt);            // (cur_* already describe the NEXT item; past the last one the values are never used)
            // first fragments of the next tile's first K-tile (skipped at steps 18-19 of this position)
            {
                const char* c2 = stage_base(s_cur);
                {
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[0][i] = EW3_LDS(c2 + a_rd[0] + i * 2048);
                }
#pragma unroll
                for (int j = 0; j < PD; ++j) bfr[j] = EW3_LDS(c2 + b_rd[0] + j * 2048);
            }
        }
    }
}

// ---- stream-K workspace: one per (device, stream) -- launches on one stream are ordered, so consecutive GEMMs may share it ----
struct SkDesc { unsigned epoch; int mode, epi, M, N, K; };                   // what a stream-K launch was (named when its hand-over times out)
constexpr int SK_DESC_RING = 1024;
struct SkWorkspace { int dev; hipStream_t stream; float* ws; unsigned* flags; unsigned epoch; SkDesc* ring; };
constexpr int SK_SLOTS = 256;
constexpr size_t SK_SLOT_FLOATS = (size_t)FM * FN * 4 * 64 * NW;          // 160 accumulators x 512 threads = 320 KB
constexpr int SK_POOL = 64;                                                  // (device, stream) pairs of the whole process
static SkWorkspace sk_pool[SK_POOL];
static int sk_pool_used = 0;
SkWorkspace* sk_pool_entry(int i) { return i < sk_pool_used ? &sk_pool[i] : nullptr; }
static std::mutex sk_mutex;                                                   // the pool is shared by every host thread
// create = false: only look the (device, stream) entry up (used while the stream is being captured into a hipGraph: allocation
// and the flag memset must not happen there -- call ew_gemm_streamk_init(stream) before the capture to get the tail inside it)
SkWorkspace* sk_workspace(hipStream_t stream, bool create) {
    std::lock_guard<std::mutex> lock(sk_mutex);
    SkWorkspace* const pool = sk_pool;
    int& used = sk_pool_used;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < used; ++i)
        if (pool[i].dev == dev && pool[i].stream == stream) return &pool[i];
    if (!create || used == SK_POOL) return nullptr;                         // pool full: those launches run the whole-tile schedule
    SkWorkspace w{dev, stream, nullptr, nullptr, 0, nullptr};
    // uncached (MTYPE UC) device memory: partials and flags cross XCDs inside one kernel, and the per-XCD L2s are only coherent
    // at kernel boundaries for ordinary allocations
    if (hipExtMallocWithFlags((void**)&w.ws, SK_SLOTS * SK_SLOT_FLOATS * sizeof(float), hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipExtMallocWithFlags((void**)&w.flags, 2048 * sizeof(unsigned), hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w.ws); return nullptr; }
    if (hipMemsetAsync(w.flags, 0, 2048 * sizeof(unsigned), stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w.flags); (void)hipFree(w.ws); return nullptr; }
    w.ring = new SkDesc[SK_DESC_RING]();
    pool[used] = w;
    return &pool[used++];
}
unsigned sk_next_epoch(SkWorkspace* w, int mode, int epi, const GemmP& p) {
    std::lock_guard<std::mutex> lock(sk_mutex);
    const unsigned e = ++w->epoch;
    w->ring[e % SK_DESC_RING] = SkDesc{e, mode, epi, p.M, p.N, p.K};
    return e;
}

}  // namespace
// 0 = every stream-K hand-over so far completed; 1 = a finisher gave up waiting (results of that launch are wrong).  Synchronises.
#if EW3_BN == 320
int ew_gemm3_sk_status_b256();
int ew_gemm3_sk_init_b256(hipStream_t s);
// Allocates the stream-K workspace of (current device, stream) for both tile instances.  Optional for eager use (the first
// launch that wants the tail allocates it); REQUIRED before capturing launches into a hipGraph, where allocation is illegal --
// a captured launch without a workspace simply runs the whole-tile schedule.
extern "C" ew_status ew_gemm_streamk_init(void* stream) {
    const bool a = sk_workspace((hipStream_t)stream, true) != nullptr;
    const bool b = ew_gemm3_sk_init_b256((hipStream_t)stream) != 0;
    if (!a || !b) { ew_set_error("ew_gemm_streamk_init: no stream-K workspace for this (device, stream): allocation failed or all 64 process-wide pool entries are in use; launches on it run the whole-tile schedule"); return EW_ERR_HIP; }
    return EW_OK;
}
extern "C" int ew_gemm_streamk_status(void) {
    const int other = ew_gemm3_sk_status_b256();
    if (other < 0) return other;
    int bad = other;
#else
int ew_gemm3_sk_init_b256(hipStream_t s) { return sk_workspace(s, true) != nullptr; }
int ew_gemm3_sk_status_b256() {
    int bad = 0;
#endif
    std::lock_guard<std::mutex> lock(sk_mutex);      // the descriptor ring is written under the same mutex by launching threads
    for (int i = 0; i < SK_POOL; ++i) {
        SkWorkspace* w = sk_pool_entry(i);
        if (!w) break;
        unsigned word[2] = {0, 0};
        if (hipMemcpy(word, w->flags + 1024, sizeof(word), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
        if (word[0]) {
            bad = 1;
            const SkDesc d = w->ring[word[1] % SK_DESC_RING];
            char msg[256];
            if (d.epoch == word[1])
                snprintf(msg, sizeof(msg), "stream-K hand-over timed out: " EW3_KERNEL_STR "<%d, %d> M=%d N=%d K=%d (stream-K launch #%u on device %d, stream %p): the output of that launch is invalid",
                         d.mode, d.epi, d.M, d.N, d.K, word[1], w->dev, (void*)w->stream);
            else
                snprintf(msg, sizeof(msg), "stream-K hand-over timed out in stream-K launch #%u on device %d, stream %p (more than %d stream-K launches ago: description no longer kept)",
                         word[1], w->dev, (void*)w->stream, SK_DESC_RING);
            ew_set_error("%s", msg);
        }
    }
    return bad;
}
namespace {

// half split of small problems: see launch3.  Smallest K it is used for:
inline int sk_half_min_k() {
    // A/B per shape at M = 7200, N = 1280 (profiles/r04_d_half_split.txt): K = 11520 conv 235 -> 222 us, K = 23040 conv 443 -> 396 us;
    // K = 5120 dense 104 -> 115 us, K = 2560 67 -> 74, temporal K = 3840 90 -> 97 (the hand-over, 2 x 320 KB per block pair, costs more
    // than the idle CUs there): on from K = 8192
    return 8192;
}
inline bool sk_half_shape(const GemmP& p, long long tiles) {
    const int mk = sk_half_min_k();
    return !(p.dbg & 4) && tiles >= 8 && 2 * tiles <= ew_cu_budget() && (2 * tiles) % 8 == 0 && (p.K / BK) % 2 == 0 && p.K >= mk;
}
template <int MODE, int EPI>
inline bool sk_half_applies(const GemmP& p, long long tiles) {
    return !(MODE == EW_A_DENSE && EPI == 23) && sk_half_shape(p, tiles);
}

template <int MODE, int EPI>
ew_status launch3(const GemmP& p, hipStream_t s) {
    GemmP q = p;
    q.tiles_m = ew_cdiv(p.M, BM);
    q.tiles_n = p.N / BN;
    q.band = ((long long)p.N * p.K * 2 > 3LL * 1024 * 1024) ? 4 : 0;
    const size_t lds = 2 * STAGE + ITEMS_BYTES;
    static std::atomic<unsigned long long> attr_mask{0};                   // per (kernel instantiation, device)
    if (ew_status st = ew_ensure_dynamic_lds((const void*)gemm3_kernel<MODE, EPI>, (int)lds, attr_mask)) return st;
    const long long tiles = (long long)q.tiles_m * q.tiles_n;
    const int NCU = ew_cu_budget();                   // 256 unless the caller runs on a CU-masked stream (ew_set_cu_budget)
    int grid = NCU;                                   // persistent: one 8-wave workgroup per CU
    if (tiles < grid) grid = (int)((tiles + 7) / 8 * 8);
    // Stream-K tail: T tiles over 256 blocks cost ceil(T/256) rounds; every C-wide output of the U-Net is 1800 / 900 / 452 tiles =
    // 7.03 / 3.52 / 1.77 rounds paid as 8 / 4 / 2.  When the loss is worth it, the last (T mod 256) + 256 tiles are cut along K
    // into 256 equal ranges instead (gemm3_kernel: contributor / finisher hand-over through the uncached workspace).
    SkP sk{nullptr, nullptr, 0u, 0, 0};
    constexpr int sk_min_k = 1280;                  // (ew_set_gemm_debug(4) forces the whole-tile schedule: the twin of the stream-K parity test)
    // Where it pays (A/B per shape on the U-Net's problems, same box): 3x3 convs at every level (-6 ... -10 %), dense / temporal
    // GEMMs with at most two tile columns and K >= 1280 (-4 ... -8 %).  With four tile columns (level 2) the blocks of an XCD
    // are out of phase along K and stop sharing the A rows and W slices in L2: +7 ... +16 % -- left on the whole-tile schedule.
    const bool sk_shape = MODE == EW_A_CONV3X3 || (q.tiles_n <= 2 && p.K >= sk_min_k);
    if (sk_shape && !(MODE == EW_A_DENSE && EPI == 23) && !(p.dbg & 4) && grid == NCU && tiles > NCU && tiles % NCU != 0) {
        const long long rounds = (tiles + NCU - 1) / NCU;
        const double loss = 1.0 - (double)tiles / ((double)NCU * rounds);
        if (loss > 0.04) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
            SkWorkspace* w = sk_workspace(s, cap == hipStreamCaptureStatusNone);
            if (w) {
                sk.dp_rounds = (int)(tiles / NCU) - 1;
                sk.tiles = (int)(tiles - (long long)NCU * sk.dp_rounds);
                sk.ws = w->ws; sk.flags = w->flags; sk.epoch = sk_next_epoch(w, MODE, EPI, p);
            }
        }
    }
    // Half split (round 4): problems with at most 128 tiles (the deepest level: M = 7200 -> 29 x 4 = 116 tiles) leave more than
    // half of the 256 CUs idle on the whole-tile schedule and used to run on generation 2's 256x160 tiles.  With a long K every
    // tile is cut into two K halves instead: 2 x tiles blocks, block 2t computes the head of tile t and finishes it with the
    // partial of block 2t+1 (same contributor / finisher hand-over as the tail split: one range = exactly half a tile).
    if (sk_half_applies<MODE, EPI>(p, tiles) && !sk.tiles) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        SkWorkspace* w = sk_workspace(s, cap == hipStreamCaptureStatusNone);
        if (w) {
            grid = (int)(2 * tiles);
            sk.dp_rounds = 0;
            sk.tiles = (int)tiles;
            sk.ws = w->ws; sk.flags = w->flags; sk.epoch = sk_next_epoch(w, MODE, EPI, p);
        }
    }
    snprintf(g_gemm_last_kernel, 64, EW3_KERNEL_STR "<%d, %d>", MODE, EPI);
    hipLaunchKernelGGL((gemm3_kernel<MODE, EPI>), dim3(grid), dim3(64 * NW), lds, s, q, sk);
    return ew_check_launch("ew_gemm_f16(gen3)");
}

// operand sets that occur in the U-Net (evoworld_amd/unet.py); any other mask runs on the smallest compiled superset
template <int MODE>
ew_status dispatch_epi3(const GemmP& p, hipStream_t s) {
    if (p.act == EW_ACT_GEGLU) {
        if constexpr (MODE == EW_A_DENSE) return launch3<MODE, 8>(p, s);
        else { ew_set_error("ew_gemm_f16: GEGLU epilogue is only built for dense mode"); return EW_ERR_UNSUPPORTED; }
    }
    const int mask = (p.rowbias ? 1 : 0) | (p.r1 ? 2 : 0) | (p.r2 ? 4 : 0);
    if (p.r1_lo || p.r2_lo || p.out_lo) {           // split-fp16 residual stream: general path with the lo companions
        // no residual operand (round 5): the row-bias enters through the accumulators' initial value, the epilogue only converts and stores
        if ((mask & 6) == 0) return launch3<MODE, 16 | 1>(p, s);
        if constexpr (MODE == EW_A_DENSE) {
            if ((mask & 4) == 0) return launch3<MODE, 16 | 3>(p, s);
            return launch3<MODE, 16 | 7>(p, s);
        } else {
            if ((mask & 5) == 0) return launch3<MODE, 16 | 2>(p, s);
            // row-bias + split output (round 3: the conv1 / temporal conv1 outputs of the resblocks, GroupNorm inputs whose fp16
            // rounding was the largest remaining storage term of the parity budget); r1 may be null (zero page)
            if ((mask & 4) == 0) return launch3<MODE, 16 | 3>(p, s);
            ew_set_error("ew_gemm_f16: conv modes carry the split residual with row-bias / r1 only (no r2)");
            return EW_ERR_UNSUPPORTED;
        }
    }
    if (mask == 0) return launch3<MODE, 0>(p, s);
    if (mask == 1) return launch3<MODE, 1>(p, s);
    if (mask == 2) return launch3<MODE, 2>(p, s);
    if constexpr (MODE == EW_A_DENSE) {
        if (mask == 3) return launch3<MODE, 3>(p, s);
        if (mask == 6 || mask == 4) return launch3<MODE, 6>(p, s);
    }
    return launch3<MODE, 7>(p, s);
}

}  // namespace

// true when generation 3 can run the problem AND is expected to be the faster choice (enough 256x320 tiles to fill the chip)
bool EW3_NAME(ew_gemm3_wants)(const GemmP& p, hipStream_t s) {
    // smallest M: the swapped-operand V^T projections (M = C = 320 / 640 / 1280 rows of W_v against N = all tokens) measured
    // 337 -> 253 us (level 0) and 192 -> 156 us (level 1) here against generation 2's 256x160 tiles (640-byte instead of 320-byte output
    // row pieces; profiles/r04_f_sweeps.txt; round 3's threshold was 1024)
    if (p.N % BN != 0 || p.M < 320) return false;
    if (BN != 320 && p.N % 320 == 0) return false;                      // the 320-wide instance takes what it can
    if ((long long)ew_cdiv(p.M, BM) * (p.N / BN) > (long long)ew_cu_budget() * (ITEMS_BYTES / 16 - 4)) return false;     // work-item table of a persistent block
    // GELU (CLIP's fc1) is only compiled into the plain dense variant: the erf code in every epilogue cost the conv variants
    // 11-28 spilled VGPRs (reloads inside the K loop, 4-10 % slower); anything else with GELU runs on generation 2
    if (p.act == EW_ACT_GELU && (p.mode != EW_A_DENSE || p.rowbias || p.r1 || p.r2 || p.out_lo)) return false;
    if (p.act == EW_ACT_SILU && (p.mode != EW_A_DENSE || p.r1 || p.r2 || p.r1_lo || p.r2_lo || p.out_lo)) return false;      // same for SiLU (round 4)
    // the epilogue addresses its row operands as uniform base + 32-bit byte offset
    const long long ld_max = max((long long)p.ld_out, max((long long)p.ld_r1, (long long)p.ld_r2));
    if ((long long)p.M * ld_max * 2 >= (1LL << 32)) return false;
    if (p.rowbias && ((long long)p.M / max(1, p.rows_per_group) + 2) * p.ld_rowbias * 2 >= (1LL << 32)) return false;
    // one tile column and a short K: 1800 tiles = 7.03 rounds over 256 CUs cost 8, and the residual-carrying epilogue is
    // store-bound anyway -- generation 2's 256x160 tiles (14.06 -> 15 rounds) measured 5-10 % faster there
    // (re-measured on the round-5 kernels: 182.90 vs 182.99 ms per forward either way, profiles/r05_g_short_rule_ab_forward.txt)
    if (p.mode == EW_A_DENSE && p.N == BN && p.K <= 1280 && (p.r1 || p.r2)) return false;
    const long long tiles = (long long)ew_cdiv(p.M, BM) * (p.N / BN);
    if (tiles * 256 >= 200LL * ew_cu_budget()) return true;      // (200 of 256 CUs busy, scaled to the CU budget)
    // fewer tiles than CUs: generation 3 only with the half split (launch3), i.e. not for the one variant compiled without it
    const bool epi23 = p.mode == EW_A_DENSE && p.r2 && (p.r1_lo || p.r2_lo || p.out_lo);        // dispatch_epi3: <0, 16|7>
    if (epi23 || p.act == EW_ACT_GEGLU || !sk_half_shape(p, tiles)) return false;
    // ... and only when launch3 can really set the split up: a (device, stream) workspace exists or may be created now (not while the
    // stream is being captured without a prior ew_gemm_streamk_init, not with the pool full / out of memory) -- otherwise the problem would
    // run whole tiles on fewer than half of the CUs, and generation 2 is the faster choice
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    return sk_workspace(s, cap == hipStreamCaptureStatusNone) != nullptr;
}

ew_status EW3_NAME(ew_gemm3_dispatch)(const GemmP& p, hipStream_t s) {
    if (p.mode == EW_A_CONV3X3) return dispatch_epi3<EW_A_CONV3X3>(p, s);
    if (p.mode == EW_A_CONVT3) return dispatch_epi3<EW_A_CONVT3>(p, s);
    return dispatch_epi3<EW_A_DENSE>(p, s);
}
