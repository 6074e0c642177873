// gemm2_f16.hip -- second-generation fused MFMA GEMM / implicit-GEMM conv for gfx950: PERSISTENT, 3-stage LDS ring.
//
// Why (profiles/r01_a, tools/bench_kernels.py): generation 1 (gemm_f16.hip: 128x160 tile, 2 blocks/CU, one
// __syncthreads per K-tile) sustains 900-1000 TF/s on large-K convs but only 200-500 TF/s on the K=320 GEMMs of
// U-Net level 0, where a block lives for 5 K-tiles and its prologue (first DMA latency) and epilogue are exposed.
// Generation 2 removes the per-tile ramp:
//   * persistent workgroups (one per CU, 8 wave64): each walks a sequence of output tiles; the K-tile stream is
//     continuous ACROSS output tiles -- the DMA of the next tile's first K-tiles flies during the current tile's
//     last MFMAs and its epilogue;
//   * 3-stage LDS ring fed by global_load_lds_dwordx4 with COUNTED s_waitcnt vmcnt(N) (never drained to 0 in the
//     stream) and a raw s_barrier: two K-tiles of DMA stay in flight across barriers;
//   * register double-buffered fragments: the ds_reads for the next MFMA half-step are issued before the current
//     half-step's MFMAs, the first fragments of K-tile v+1 are read right after the barrier that publishes it --
//     no LDS-latency bubble at the K-tile boundary, one barrier per K-tile;
//   * swapped MFMA operands (D = W_frag x A_frag^T): every lane ends up with 4 consecutive output columns of one row; the
//     epilogue (bias / row-bias / SiLU / GEGLU / two residuals) goes through a wave-private fp32 LDS patch living in the ring
//     slot just consumed and issues 16-byte row-major stores, with the row operands requested one store step ahead;
//   * the DMA pieces of a K-tile are issued between the MFMAs of two half-steps instead of back to back after the barrier.
// Generation 3 (gemm3_f16.hip, 256x320 tile) is the default where it applies; this generation handles the other shapes.
// Same argument block, same addressing modes (dense / conv3x3 / temporal 3-tap, dual source, zero page) as gen 1.
#include "gemm_common.h"
#include <type_traits>


extern char g_gemm_last_kernel[64];

namespace {

#define EW_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// gfx9 s_waitcnt simm16: vmcnt[3:0]=bits3:0, expcnt=bits6:4, lgkmcnt=bits11:8, vmcnt[5:4]=bits15:14.  The BUILTIN form is
// used for lgkmcnt so that hipcc's own waitcnt model knows the LDS queue is empty (an inline-asm wait is opaque to it).
#define EW_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define EW_COMPILER_FENCE() asm volatile("" ::: "memory")

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI: compile-time epilogue operand set -- bit0 row-bias, bit1 residual r1, bit2 residual r2, bit3 GEGLU.  An operand
// that is compiled in but absent at run time is read from the zero page with stride 0 (so a superset kernel is always valid).
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int MODE, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N == 4 && BM == 256) ? 1 : 2) void gemm2_kernel(const GemmP p) {
    constexpr int NW = WAVES_M * WAVES_N;                  // 8 waves, 1 workgroup/CU  -or-  4 waves, 2 workgroups/CU
    constexpr int BK = 64;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int A_GROUPS = BM / 8, B_GROUPS = BN / 8;
    constexpr int GA = A_GROUPS / NW;                      // A row-groups per wave (A_GROUPS is a multiple of NW)
    constexpr int GB = (B_GROUPS + NW - 1) / NW;           // W row-groups per wave (the last may be partial)
    constexpr int GB_FULL = B_GROUPS / NW;                 // W row-groups every wave owns
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int A_BYTES = BM * 128;
    static_assert((NW == 8 || NW == 4) && (WM == 64 || WM == 128) && A_GROUPS % NW == 0 && (NSTAGE == 2 || NSTAGE == 3), "config");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const bool has_tail = (GB > GB_FULL) && (wave + NW * GB_FULL < B_GROUPS);   // owns the partial last W group
    const int n_ld = GA + GB_FULL + (has_tail ? 1 : 0);                         // DMA instructions per K-tile (this wave)

    // ---- tile sequence of this persistent block: step i -> tile id i*G + (b%8)*(G/8) + b/8  (XCD-contiguous chunks)
    const int G = gridDim.x;
    const int total_tiles = p.tiles_m * p.tiles_n;
    const int seq0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int n_my = seq0 < total_tiles ? (total_tiles - 1 - seq0) / G + 1 : 0;
    const int C = p.c1 + p.c2;
    const int nk = p.K / BK;
    const int V = n_my * nk;                                                     // virtual K-tile stream length
    if (V == 0) return;

    const int srow = lane >> 3;
    const int slot = (lane & 7) ^ srow;

    // ---------------- loader state (runs NSTAGE K-tiles ahead of the MFMA stream, across output tiles) ----------------
    // Per A row of this lane: element offset of the CENTRE tap in each source tensor (+ the lane's 16-byte slot), a tap
    // validity mask, and for the nearest-x2-upsampled conv a 2-bit (dy, dx) code per tap.  A K-tile's source address is then
    // centre + a wave-uniform tap delta: ~5 VALU per row per K-tile, no re-derivation of (img, y, x) inside the stream
    // (an ablation showed the loader's address arithmetic, not DMA bandwidth, was costing ~30 % of the main loop).
    int ld_i = 0, ld_kt = 0, ld_tap = 0, ld_cc = 0;
    int a_ctr[GA];                         // centre-tap pixel (row) index in the source tensors
    int a_mask[GA];                        // bits 0..8 tap validity, bits 16..27 upsample (dy,dx) codes
    const f16* b_ptr[GB];
    constexpr int NTAP = MODE == EW_A_CONV3X3 ? 9 : (MODE == EW_A_CONVT3 ? 3 : 1);

    auto loader_new_tile = [&]() __attribute__((always_inline)) {
        const int id = ld_i * G + seq0;
        const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            int m = m0 + (wave + NW * i) * 8 + srow;
            m = m < p.M ? m : p.M - 1;
            long long ctr;
            int mask = 1, dcode = 0;
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
                    ctr = ((long long)img * p.h_in + sy) * p.w_in + sx;
                } else {
                    ctr = ((long long)img * p.h_in + cy) * p.w_in + cx;
                }
            } else if constexpr (MODE == EW_A_CONVT3) {
                const int tp = p.tT * p.tP;
                const int bb = m / tp, rem = m - bb * tp;
                const int t = rem / p.tP, x = rem - t * p.tP;
                mask = (t > 0 ? 1 : 0) | 2 | (t + 1 < p.tT ? 4 : 0);
                ctr = ((long long)bb * p.tT + t) * p.tP + x;
            } else {
                ctr = m;
            }
            a_mask[i] = mask | (dcode << 16);
            a_ctr[i] = (int)ctr;
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            int n = n0 + (wave + NW * j) * 8 + srow;
            n = n < p.N ? n : p.N - 1;
            b_ptr[j] = p.w + (size_t)n * p.K + slot * 8;
        }
        ld_kt = 0; ld_tap = 0; ld_cc = 0;
    };

    // Staging of one K-tile = stage_begin (wave-uniform source selection, advances the stream counters) + NP DMA pieces.
    // In the steady state the pieces are INTERLEAVED with the MFMAs (mma_il below): issuing an LDS-DMA instruction costs the
    // issuing wave ~60-180 cycles, and with all pieces issued back to back after the barrier both waves of a SIMD sat in
    // DMA issue at the same time with the MFMA pipe idle (exp10: ~650 of ~2500 cycles per K-tile).
    constexpr int NP = GA + GB;                                             // pieces per K-tile (the last W piece may be absent)
    constexpr int P0 = NP / 2;                                              // pieces [0,P0) ride half-step 1, [P0,NP) the next half-step 0
    const f16* st_base = p.a;
    const f16* st_zp = p.zero_page + slot * 8;
    long long st_dl = 0;
    bool st_second = false;
    int st_tap = 0, st_ld = 0, st_ch = 0;
    size_t st_koff = 0;
    char* st_buf = smem;
    auto stage_begin = [&](char* buf) __attribute__((always_inline)) {
        if (ld_kt == 0) loader_new_tile();
        st_buf = buf;
        st_tap = ld_tap;
        const int cc = ld_cc;
        st_second = cc >= p.c1;
        st_base = st_second ? p.a2 : p.a;
        st_ld = st_second ? p.lda2 : p.lda;
        st_ch = st_second ? cc - p.c1 : cc;
        int dpix = 0;                                                       // wave-uniform tap delta in pixels
        if constexpr (MODE == EW_A_CONV3X3) dpix = (st_tap / 3 - 1) * p.w_in + (st_tap % 3 - 1);
        else if constexpr (MODE == EW_A_CONVT3) dpix = (st_tap - 1) * p.tP;
        st_dl = (long long)dpix * st_ld + st_ch;
        st_koff = (size_t)ld_kt * BK;
        // K order is CHANNEL-CHUNK major, tap minor: the taps of one 64-channel chunk re-read (shifted) the same input
        // lines, so the chunk's footprint (~66 KB per workgroup) is fetched from HBM/MALL once and re-hit in L2 for the
        // other taps; tap-major order re-fetched the whole C-wide footprint (10 MB per XCD > 4 MB L2) for every tap.
        if (++ld_tap == NTAP) { ld_tap = 0; ld_cc += BK; }
        if (++ld_kt == nk) { ld_kt = 0; ++ld_i; }
    };
    auto stage_piece = [&](int k) __attribute__((always_inline)) {     // k is a compile-time constant after unrolling
        if (k < GA) {
            const int i = k;
            // element offset = centre row * row stride + (tap delta + channel + this lane's 16-byte slot): one v_mad_u64_u32
            const f16* src = st_base + ((long long)a_ctr[i] * st_ld + (st_dl + slot * 8));
            if constexpr (MODE == EW_A_CONV3X3) {
                if (p.upsample) {                                           // per-row deltas (nearest-x2 source coordinates)
                    const int dc = a_mask[i] >> 16;
                    const int dy = ((dc >> (2 * (st_tap / 3))) & 3) - 1, dx = ((dc >> (6 + 2 * (st_tap % 3))) & 3) - 1;
                    src = st_base + ((long long)(a_ctr[i] + dy * p.w_in + dx) * st_ld + (st_ch + slot * 8));
                }
            }
            if constexpr (MODE != EW_A_DENSE) src = ((a_mask[i] >> st_tap) & 1) ? src : st_zp;
            glds16(src, st_buf + (wave + NW * i) * 1024);
        } else {
            const int j = k - GA;
            if (j < GB_FULL || has_tail) glds16(b_ptr[j] + st_koff, st_buf + A_BYTES + (wave + NW * j) * 1024);
        }
    };
    auto stage = [&](char* buf) __attribute__((always_inline)) {       // whole K-tile at once (prologue, and the DMA deferred past an epilogue)
        stage_begin(buf);
#pragma unroll
        for (int k = 0; k < NP; ++k) stage_piece(k);
    };

    // Output stores issued by this wave in one FULL-tile epilogue (exact instruction count: every store executes).
    constexpr int NST = FM * ((((EPI & 8) ? 16 * (WN / 16) : 16 * (WN / 8)) + 63) / 64);   // 16-byte row-major stores
    constexpr int NST2 = (EPI & 16) ? 2 * NST : NST;                  // split-fp16 residual stream: hi and lo stores
    const bool lo_out = (EPI & 16) && p.out_lo != nullptr;
    static_assert(GA + GB + 1 + NST2 < 64, "vmcnt is a 6-bit counter");
    // own DMA of the NEXT K-tile landed; the newest K-tile (n_ld DMA instructions) stays in flight.  `stores_behind`:
    // the NST output stores of the tile just finished were issued AFTER the DMA we wait for -- vmcnt counts in issue
    // order, so they are allowed to stay in flight too and the wave does not stall on store acknowledgements.
    auto wait_landed = [&](bool more_in_flight, bool stores_behind) __attribute__((always_inline)) {
        if constexpr (NSTAGE == 2) { wait_vmcnt<0>(); return; }         // nothing newer than the tile we wait for
        else {
            if (!more_in_flight) { wait_vmcnt<0>(); return; }
            if constexpr (GB > GB_FULL) {
                if (has_tail) { if (stores_behind) { if (lo_out) wait_vmcnt<GA + GB_FULL + 1 + NST2>(); else wait_vmcnt<GA + GB_FULL + 1 + NST>(); } else wait_vmcnt<GA + GB_FULL + 1>(); }
                else { if (stores_behind) { if (lo_out) wait_vmcnt<GA + GB_FULL + NST2>(); else wait_vmcnt<GA + GB_FULL + NST>(); } else wait_vmcnt<GA + GB_FULL>(); }
            } else {
                if (stores_behind) { if (lo_out) wait_vmcnt<GA + GB_FULL + NST2>(); else wait_vmcnt<GA + GB_FULL + NST>(); } else wait_vmcnt<GA + GB_FULL>();
            }
        }
    };

    // ---------------- fragment geometry ----------------
    const int frow = lane & 15, fks = lane >> 4, sw = frow & 7;
    int a_off[FM], b_off[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a_off[i] = (wm * WM + i * 16 + frow) * 128;
#pragma unroll
    for (int j = 0; j < FN; ++j) b_off[j] = A_BYTES + (wn * WN + j * 16 + frow) * 128;
    const int so0 = ((0 * 4 + fks) ^ sw) << 4, so1 = ((1 * 4 + fks) ^ sw) << 4;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f16x8 af0[FM], bf0[FN], af1[FM], bf1[FN];
    auto read_frags = [&](const char* buf, int so, f16x8 (&af)[FM], f16x8 (&bf)[FN]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(buf + b_off[j] + so);
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(buf + a_off[i] + so);
    };
    auto mma = [&](const f16x8 (&af)[FM], const f16x8 (&bf)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);   // D[n][m]
    };

    // MFMAs of one half-step with DMA pieces [LO,HI) of the pending stage spread evenly between them
    auto mma_il = [&](const f16x8 (&af)[FM], const f16x8 (&bf)[FN], bool on, auto lo_tag, auto hi_tag) __attribute__((always_inline)) {
        constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value, NQ = HI - LO, NM = FM * FN;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);   // D[n][m]
                const int idx = i * FN + j;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (idx == ((q + 1) * NM) / (NQ + 1) - 1) {
                        if (on) {
                            __builtin_amdgcn_sched_barrier(0);
                            stage_piece(LO + q);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
    };

    // ---------------- prologue: NSTAGE K-tiles requested, first fragments in registers ----------------
    // Ring rule: barrier(v) publishes stream position v+1 and frees slot v%NSTAGE; the DMA of position v+NSTAGE is issued
    // right after it -- except when position v ends an output tile: then the freed slot first hosts the epilogue patches
    // and the DMA is issued after the epilogue's closing barrier.
    int staged = 0;                                  // stream positions whose DMA has been issued
    for (; staged < NSTAGE && staged < V; ++staged) stage(smem + staged * STAGE);
    if (V > 1) {
        if constexpr (NSTAGE == 3) { if (V > 2) wait_landed(true, false); else { wait_vmcnt<0>(); } }
        else wait_vmcnt<0>();
    } else {
        wait_vmcnt<0>();
    }
    if constexpr (NSTAGE == 2) { /* both tiles waited: simplest, once per launch */ }
    EW_COMPILER_FENCE();
    __builtin_amdgcn_s_barrier();
    EW_COMPILER_FENCE();
    read_frags(smem, so0, af0, bf0);

    int cur_i = 0, cur_kt = 0;
    bool stores_behind = false;                      // a full-tile epilogue's stores are newer than the DMA waited next
    int s_cur = 0;                                   // ring slot of stream position v
    bool pend = false;                               // pieces [P0,NP) of the newest stage still to be issued
    for (int v = 0; v < V; ++v) {
        const int s_nxt = s_cur == NSTAGE - 1 ? 0 : s_cur + 1;
        const char* cur = smem + s_cur * STAGE;
        const bool tile_end = cur_kt == nk - 1;
        // ---- half-step 0: MFMA on k[0,32), fetch fragments of k[32,64)
        EW_WAIT_LGKM0();   // af0/bf0 (read one half-step ago) have landed: free, and it lets the MFMAs below start
                           // without waiting for the reads issued next (hipcc otherwise emits lgkmcnt(0) after them)
        read_frags(cur, so1, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads AHEAD of the MFMAs (hipcc otherwise sinks them to the end)
        mma_il(af0, bf0, pend, std::integral_constant<int, P0>{}, std::integral_constant<int, NP>{});
        pend = false;
        // ---- publish K-tile v+1.  Unconditional (also on the last position, where the fragments read from the ring are
        // stale and never used): a conditional here makes hipcc put a conservative lgkmcnt(0) at the join, in front of
        // the half-step-1 MFMAs, which would expose the LDS latency of the reads just issued.
        wait_landed(v + 2 < staged, stores_behind);
        stores_behind = false;
        EW_WAIT_LGKM0();
        EW_COMPILER_FENCE();
        __builtin_amdgcn_s_barrier();
        EW_COMPILER_FENCE();
        read_frags(smem + s_nxt * STAGE, so0, af0, bf0);
        const bool st_now = !tile_end && staged < V;                             // slot of v is free from here on
        if (st_now) {
            ++staged;
            stage_begin(smem + s_cur * STAGE);
            pend = true;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- half-step 1
        mma_il(af1, bf1, st_now, std::integral_constant<int, 0>{}, std::integral_constant<int, P0>{});
        const int s_prev = s_cur;
        s_cur = s_nxt;
        if (++cur_kt == nk) {
            // ------------------------- epilogue of output tile cur_i (registers only) -------------------------
            cur_kt = 0;
            const int id = cur_i * G + seq0;
            ++cur_i;
            const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
            // every store instruction of this wave executes iff its whole wave tile is inside the matrix
            stores_behind = (tm * BM + wm * WM + WM <= p.M) && (tn * BN + wn * WN + WN <= p.N);
            // Epilogue through a wave-private LDS patch living in the ring slot of the K-tile just consumed (free since
            // barrier(v); protected from the next DMA by the barrier at the end): accumulators (D[n][m] layout = 4
            // consecutive columns per lane) -> ds_write_b128 -> read back ROW-major, 8 columns per lane -> every global
            // access of the fused epilogue (bias, row-bias, residuals, output) is a 16-byte piece of a contiguous row
            // segment (WN*2 bytes), instead of 8-byte pieces of 32-byte segments straight from the MFMA layout
            // (measured: 2.7 TB/s store rate, the dominant cost of the K=320 GEMMs).
            constexpr int LDP = WN + 4;                            // patch row stride (floats)
            float* patch = (float*)(smem + s_prev * STAGE) + wave * (16 * LDP);
            const int m_w0 = tm * BM + wm * WM, n_w0 = tn * BN + wn * WN;
            auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;   // FULL: no per-access guards -> exact store count
                // an operand compiled in (EPI) but absent at run time reads the zero page with stride 0
                const f16* bp = p.bias ? p.bias : p.zero_page;
                const f16* rbp = p.rowbias ? p.rowbias : p.zero_page;
                const f16* r1p = p.r1 ? p.r1 : p.zero_page;
                const f16* r2p = p.r2 ? p.r2 : p.zero_page;
                const int mbias = p.bias ? 1 : 0, mrb = p.rowbias ? 1 : 0, m1 = p.r1 ? 1 : 0, m2 = p.r2 ? 1 : 0;
                const int ldrb = p.rowbias ? p.ld_rowbias : 0, ld1 = p.r1 ? p.ld_r1 : 0, ld2 = p.r2 ? p.ld_r2 : 0;
                // vmcnt is an IN-ORDER counter: a load issued after a store cannot be waited for without also draining that
                // store.  So (a) bias vectors (column-only) are loaded once, before any store; (b) the row operands of step
                // s+1 (row-bias, residuals) are requested BEFORE the store of step s.  With loads interleaved after each store
                // the epilogue serialised into ~12 store round trips per tile (measured: epilogue = main loop at K=320).
                if constexpr ((EPI & 8) == 0) {
                    constexpr bool RB = EPI & 1, R1 = EPI & 2, R2 = EPI & 4, LO = EPI & 16;
                    const int8_t* r1lp = p.r1_lo ? p.r1_lo : (const int8_t*)p.zero_page;      // split residual stream: lo8 companions
                    const int8_t* r2lp = p.r2_lo ? p.r2_lo : (const int8_t*)p.zero_page;
                    const int m1l = p.r1_lo ? 1 : 0, m2l = p.r2_lo ? 1 : 0;
                    const int ld1l = p.r1_lo ? p.ld_r1 : 0, ld2l = p.r2_lo ? p.ld_r2 : 0;
                    constexpr int VPR = WN / 8;                 // 16-byte output vectors per row
                    constexpr int ITERS = (16 * VPR + 63) / 64;
                    constexpr int NS = FM * ITERS;
                    int rowv[ITERS], c8v[ITERS];
                    f16x8 bvv[ITERS];
                    auto is_live = [&](int it) { return (16 * VPR) % 64 == 0 || it * 64 + lane < 16 * VPR; };
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int idx = it * 64 + lane;
                        rowv[it] = is_live(it) ? idx / VPR : 0;
                        c8v[it] = is_live(it) ? (idx - rowv[it] * VPR) * 8 : 0;
                        const int n = n_w0 + c8v[it];
                        bvv[it] = *(const f16x8*)(bp + ((FULL || n + 8 <= p.N) ? n : 0) * mbias);
                    }
                    f16x8 rbv[2], q1v[2], q2v[2];
                    u32x2 q1l[2], q2l[2];
                    auto fetch = [&](int i, int it, int set) {
                        const int m = m_w0 + i * 16 + rowv[it], n = n_w0 + c8v[it];
                        const int mc = FULL ? m : min(m, p.M - 1), nc = (FULL || n + 8 <= p.N) ? n : 0;
                        if constexpr (RB) rbv[set] = *(const f16x8*)(rbp + (size_t)(mc / p.rows_per_group) * ldrb + nc * mrb);
                        if constexpr (R1) q1v[set] = *(const f16x8*)(r1p + (size_t)mc * ld1 + nc * m1);
                        if constexpr (R2) q2v[set] = *(const f16x8*)(r2p + (size_t)mc * ld2 + nc * m2);
                        if constexpr (R1 && LO) q1l[set] = *(const u32x2*)(r1lp + (size_t)mc * ld1l + nc * m1l);
                        if constexpr (R2 && LO) q2l[set] = *(const u32x2*)(r2lp + (size_t)mc * ld2l + nc * m2l);
                    };
                    fetch(0, 0, 0);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            *(f32x4*)(patch + frow * LDP + j * 16 + fks * 4) = acc[i][j];
                            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int it = 0; it < ITERS; ++it) {
                            const int set = (i * ITERS + it) & 1;
                            if (it + 1 < ITERS) fetch(i, it + 1, set ^ 1);
                            else if (i + 1 < FM) fetch(i + 1, 0, set ^ 1);
                            const int row = rowv[it], c8 = c8v[it];
                            const int m = m_w0 + i * 16 + row, n = n_w0 + c8;
                            const f32x4 lo = *(const f32x4*)(patch + row * LDP + c8);
                            const f32x4 hi = *(const f32x4*)(patch + row * LDP + c8 + 4);
                            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            f16x8 o;
                            int s8[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float x = v[e] + (float)bvv[it][e];
                                if constexpr (RB) x += (float)rbv[set][e];
                                if (p.act == EW_ACT_SILU) x = ew_silu(x);
                                else if (p.act == EW_ACT_GELU) x = ew_gelu(x);
                                x *= p.c_acc;
                                if constexpr (R1 && LO) x += p.c_r1 * ew_split_dec(q1v[set][e], ew_sbyte(q1l[set][e >> 2], e & 3));
                                else if constexpr (R1) x += p.c_r1 * (float)q1v[set][e];
                                if constexpr (R2 && LO) x += p.c_r2 * ew_split_dec(q2v[set][e], ew_sbyte(q2l[set][e >> 2], e & 3));
                                else if constexpr (R2) x += p.c_r2 * (float)q2v[set][e];
                                o[e] = (f16)x;
                                if constexpr (LO) s8[e] = ew_split_enc(x, o[e]);
                            }
                            if (FULL ? is_live(it) : (is_live(it) && m < p.M && n + 8 <= p.N)) {
                                *(f16x8*)(p.out + (size_t)m * p.ld_out + n) = o;
                                if constexpr (LO) {
                                    if (p.out_lo)
                                        *(u32x2*)(p.out_lo + (size_t)m * p.ld_out + n) =
                                            (u32x2){ew_pack4(s8[0], s8[1], s8[2], s8[3]), ew_pack4(s8[4], s8[5], s8[6], s8[7])};
                                }
                            } else if (!FULL && is_live(it) && m < p.M) {
#pragma unroll
                                for (int e = 0; e < 8; ++e)                     // ragged N edge (e.g. conv_out N=4)
                                    if (n + e < p.N) {
                                        p.out[(size_t)m * p.ld_out + n + e] = o[e];
                                        if constexpr (LO) {
                                            if (p.out_lo) p.out_lo[(size_t)m * p.ld_out + n + e] = (int8_t)s8[e];
                                        }
                                    }
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                } else {
                    // GEGLU: staged column blocks of 32 = [16 value | 16 gate] -> fragment 2q holds values, 2q+1 the gates of
                    // the SAME (row, column) positions in the SAME lane: value*gelu(gate) is formed in registers, only the
                    // WN/2 output columns go through the (half-width) patch, and a row's output is one contiguous run.
                    constexpr int WO = WN / 2, LDO = WO + 4;
                    constexpr int VPR = WO / 8;                 // 16-byte output vectors per row
                    constexpr int ITERS = (16 * VPR + 63) / 64;
                    f32x4 bq[FN];
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const int n = n_w0 + j * 16 + fks * 4;
                        const f16x4 b4 = *(const f16x4*)(bp + ((FULL || n < p.N) ? n : 0) * mbias);
                        bq[j] = (f32x4){(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
                    }
                    auto is_live = [&](int it) { return (16 * VPR) % 64 == 0 || it * 64 + lane < 16 * VPR; };
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
#pragma unroll
                        for (int q = 0; q < FN / 2; ++q) {
                            const f32x4 vv = acc[i][2 * q] + bq[2 * q], gg = acc[i][2 * q + 1] + bq[2 * q + 1];
                            const f32x2 o01 = ew_vgelu2((f32x2){vv[0], vv[1]}, (f32x2){gg[0], gg[1]});
                            const f32x2 o23 = ew_vgelu2((f32x2){vv[2], vv[3]}, (f32x2){gg[2], gg[3]});
                            const f32x4 o4 = {o01[0], o01[1], o23[0], o23[1]};
                            *(f32x4*)(patch + frow * LDO + q * 16 + fks * 4) = o4;
                            acc[i][2 * q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            acc[i][2 * q + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int it = 0; it < ITERS; ++it) {
                            const int idx = it * 64 + lane;
                            const int row = is_live(it) ? idx / VPR : 0, c8 = is_live(it) ? (idx - row * VPR) * 8 : 0;
                            const int m = m_w0 + i * 16 + row;
                            const f32x4 lo = *(const f32x4*)(patch + row * LDO + c8), hi = *(const f32x4*)(patch + row * LDO + c8 + 4);
                            const f16x8 o = {(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
                            const int no = (n_w0 >> 1) + c8;
                            if (FULL ? is_live(it) : (is_live(it) && m < p.M && 2 * no < p.N))
                                *(f16x8*)(p.out + (size_t)m * p.ld_out + no) = o;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            };
            {
                if (stores_behind) epilogue(std::true_type{}); else epilogue(std::false_type{});
                // the patch lives in the ring slot the NEXT DMA (stage at the top of the next position) will overwrite
                EW_WAIT_LGKM0();
                EW_COMPILER_FENCE();
                __builtin_amdgcn_s_barrier();
                EW_COMPILER_FENCE();
            }
            if (staged < V) { stage(smem + s_prev * STAGE); ++staged; }   // the DMA deferred at this tile's last barrier
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int MODE, int EPI>
ew_status launch2(const GemmP& p, hipStream_t s) {
    constexpr int NW = WAVES_M * WAVES_N;
    GemmP q = p;
    q.tiles_m = ew_cdiv(p.M, BM);
    q.tiles_n = ew_cdiv(p.N, BN);
    const size_t lds = NSTAGE * (BM + BN) * 128;
    static std::atomic<unsigned long long> attr_mask{0};                   // per (kernel instantiation, device)
    if (ew_status st = ew_ensure_dynamic_lds((const void*)gemm2_kernel<BM, BN, WAVES_M, WAVES_N, NSTAGE, MODE, EPI>, (int)lds, attr_mask)) return st;
    const long long tiles = (long long)q.tiles_m * q.tiles_n;
    if (tiles <= 0 || tiles > 0x7fffffffLL) { ew_set_error("ew_gemm_f16: bad grid"); return EW_ERR_INVALID_ARG; }
    int grid = (NW == 8 || BM == 256) ? ew_cu_budget() : 2 * ew_cu_budget();                   // persistent: 1 x 8-wave or 2 x 4-wave workgroups per CU (256 CUs)
    if (tiles < grid) grid = (int)((tiles + 7) / 8 * 8);
    snprintf(g_gemm_last_kernel, 64, "gemm2_kernel<%d, %d, %d, %d, %d, %d, %d>", BM, BN, WAVES_M, WAVES_N, NSTAGE, MODE, EPI);
    hipLaunchKernelGGL((gemm2_kernel<BM, BN, WAVES_M, WAVES_N, NSTAGE, MODE, EPI>), dim3(grid), dim3(64 * NW), lds, s, q);
    return ew_check_launch("ew_gemm_f16(gen2)");
}

// Tile / workgroup shape: 8-wave workgroups, 3-stage ring, 256x160 where N is a multiple of 160 or at most 160, else (and for GEGLU) 128x256.
// (Measured and dropped: 2 workgroups of 4 waves per CU with 128x160 / 128x128 tiles and a 2-stage ring, with and without a
// start-phase offset; chip-wide start staggering; 4 waves x 512 VGPRs; a 256x256 / 2-stage GEGLU tile -- DESIGN.md 3.1.)
template <int MODE, int EPI>
ew_status dispatch_tile(const GemmP& p, hipStream_t s) {
    if constexpr (EPI & 8) {
        return launch2<128, 256, 2, 4, 3, MODE, EPI>(p, s);
    } else {
        // N <= 160 (the VAE's 128-channel convs at full resolution): one 160-wide tile column wastes 20 % of it, a 256-wide one half
        if (p.N % 160 == 0 || p.N <= 160) return launch2<256, 160, 4, 2, 3, MODE, EPI>(p, s);
        return launch2<128, 256, 2, 4, 3, MODE, EPI>(p, s);
    }
}

// operand sets that occur in the U-Net (evoworld_amd/unet.py); any other mask runs on the smallest compiled superset
template <int MODE>
ew_status dispatch_epi(const GemmP& p, hipStream_t s) {
    if (p.act == EW_ACT_GEGLU) {
        if constexpr (MODE == EW_A_DENSE) return dispatch_tile<MODE, 8>(p, s);
        else { ew_set_error("ew_gemm_f16: GEGLU epilogue is only built for dense mode"); return EW_ERR_UNSUPPORTED; }
    }
    const int mask = (p.rowbias ? 1 : 0) | (p.r1 ? 2 : 0) | (p.r2 ? 4 : 0);
    if (p.r1_lo || p.r2_lo || p.out_lo) {           // split-fp16 residual stream
        if constexpr (MODE == EW_A_DENSE) {
            if ((mask & 4) == 0) return dispatch_tile<MODE, 16 | 3>(p, s);
            return dispatch_tile<MODE, 16 | 7>(p, s);
        } else {
            if ((mask & 5) == 0) return dispatch_tile<MODE, 16 | 2>(p, s);
            // row-bias + split output (round 3: the conv1 / temporal conv1 outputs of the resblocks, GroupNorm inputs whose fp16
            // rounding was the largest remaining storage term of the parity budget); r1 may be null (zero page)
            if ((mask & 4) == 0) return dispatch_tile<MODE, 16 | 3>(p, s);
            ew_set_error("ew_gemm_f16: conv modes carry the split residual with row-bias / r1 only (no r2)");
            return EW_ERR_UNSUPPORTED;
        }
    }
    if (mask == 0) return dispatch_tile<MODE, 0>(p, s);
    if (mask == 1) return dispatch_tile<MODE, 1>(p, s);
    if (mask == 2) return dispatch_tile<MODE, 2>(p, s);
    if constexpr (MODE == EW_A_DENSE) {
        if (mask == 3) return dispatch_tile<MODE, 3>(p, s);
        if (mask == 6 || mask == 4) return dispatch_tile<MODE, 6>(p, s);
    }
    return dispatch_tile<MODE, 7>(p, s);
}

}  // namespace


ew_status ew_gemm2_dispatch(const GemmP& p, hipStream_t s) {
    if (p.mode == EW_A_CONV3X3) return dispatch_epi<EW_A_CONV3X3>(p, s);
    if (p.mode == EW_A_CONVT3) return dispatch_epi<EW_A_CONVT3>(p, s);
    return dispatch_epi<EW_A_DENSE>(p, s);
}
