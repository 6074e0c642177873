// gemm4_f16.hip -- generation 4 (round 6): dense GEMM in the shape of the vendor's large-K solutions (profiles/r04_a_hipblaslt_solutions.txt):
// 256 x 256 x 64 tile, FOUR waves (one per SIMD, the whole 512-register file each), 128 x 128 wave tile = 16 blocks of v_mfma_f32_32x32x16_f16 whose
// 256 accumulator registers live in AccVGPRs (inline-asm MFMAs with "+a" operands: hipcc keeps builtin MFMA results in arch VGPRs), operands
// global -> VGPR -> ds_write -> LDS with the global reads TWO K-tiles ahead of their use (128 VGPRs of loads in flight per lane; generation 3's
// LDS-DMA feed keeps one 72 KB stage in flight per CU, which is what bounds it when A streams from HBM: DESIGN.md 3.3), LDS double-buffered, one
// barrier per K-tile, a continuous K-tile stream across the output tiles of a persistent workgroup (the next tile's first two K-tiles are in flight
// while the epilogue runs).
// Scope: the bias-only / GEGLU dense shapes with N % 256 == 0 (the GEGLU up-projections of levels 1-2: 30 launches, 19.6 ms per forward), plain fp16
// output.  out = c_acc * act(A W^T + bias), act in {none, GEGLU}.  Reached through ew_gemm_f16 with ew_set_gemm_generation(4) (A/B) -- see DESIGN.md 3.6
// for what it measured.
#include "gemm_common.h"
#include "gemm4_acc.inc"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NW = 4;
constexpr int STAGE = (BM + BN) * 128;            // 64 KB: rows 0..255 = A tile (activations), rows 256..511 = W tile, 128 B (64 k) per row
constexpr int NPIECE = (BM + BN) * 8 / (64 * NW); // 16-byte pieces per thread and K-tile: 16

__device__ __forceinline__ int swz4(int row) { return (row ^ (row >> 3)) & 7; }

// block k = 4 i + j (n block i, m block j) lives in a[16 k : 16 k + 15]
#define G4_ALL(M, wf, af)                                                                                                                     \
    M##_0(wf[0], af[0]); M##_1(wf[0], af[1]); M##_2(wf[0], af[2]); M##_3(wf[0], af[3]); M##_4(wf[1], af[0]); M##_5(wf[1], af[1]);              \
    M##_6(wf[1], af[2]); M##_7(wf[1], af[3]); M##_8(wf[2], af[0]); M##_9(wf[2], af[1]); M##_10(wf[2], af[2]); M##_11(wf[2], af[3]);            \
    M##_12(wf[3], af[0]); M##_13(wf[3], af[1]); M##_14(wf[3], af[2]); M##_15(wf[3], af[3])

__device__ __forceinline__ void tile_coords4(int id, int tiles_m, int tiles_n, int band, int& tm, int& tn) {
    if (band <= 0 || tiles_n <= band) { tm = id / tiles_n; tn = id - tm * tiles_n; return; }
    const int nb = (tiles_n + band - 1) / band;
    const int per_band = band * tiles_m;
    const int k = min(id / per_band, nb - 1);
    const int r = id - k * per_band;
    const int w = k == nb - 1 ? tiles_n - k * band : band;
    tm = r / w;
    tn = k * band + (r - tm * w);
}

template <int ACT>      // 0: none, 2: GEGLU (value / gate rows interleaved in blocks of 16 by the host pack, as for generation 3)
__global__ __launch_bounds__(64 * NW, 1) void gemm4_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lq = lane & 31, lh = lane >> 5;

    const int G = gridDim.x;
    const int total = p.tiles_m * p.tiles_n;
    const int seq0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);      // XCD-contiguous chunks of the tile order
    const int n_my = seq0 < total ? (total - 1 - seq0) / G + 1 : 0;
    if (n_my == 0) return;
    const int nk = p.K / BK;
    const int V = n_my * nk;                                               // K-tile stream length of this workgroup

    // ---- A loader: thread t owns pieces (row t/8 + 32 j, 16-byte slot t & 7), j = 0..7 of the 256 x 64 A tile; global -> VGPR two K-tiles ahead
    const int prow = tid >> 3, pslot = tid & 7;
    int a_off[8];                                                          // element offset of A row (tm * 256 + prow + 32 j, clamped) + slot (< 2^31: checked by the host)
    int ld_tile = 0, ld_kt = 0;                                            // stream position of the NEXT A load: tile index in my list, K-tile
    f16x8 ldr[2][8];
    auto issue_loads = [&](auto set_tag) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
        if (ld_kt == 0) {
            int tm, tn;
            tile_coords4(seq0 + ld_tile * G, p.tiles_m, p.tiles_n, p.band, tm, tn);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int m = tm * BM + prow + 32 * j;
                m = m < p.M ? m : p.M - 1;
                a_off[j] = m * p.lda + pslot * 8;
            }
        }
        const f16* ap = p.a + ld_kt * BK;
#pragma unroll
        for (int j = 0; j < 8; ++j) ldr[SET][j] = *(const f16x8*)(ap + a_off[j]);
        if (++ld_kt == nk) { ld_kt = 0; ++ld_tile; }
    };
    // the same in two parts for the hand-placed main loop: a_prep() (tile change + base pointer), then A_PIECE(SET, j) one load at a time
    const f16* a_kp = p.a;
    auto a_prep = [&]() __attribute__((always_inline)) {
        if (ld_kt == 0) {
            int tm, tn;
            tile_coords4(seq0 + ld_tile * G, p.tiles_m, p.tiles_n, p.band, tm, tn);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int m = tm * BM + prow + 32 * j;
                m = m < p.M ? m : p.M - 1;
                a_off[j] = m * p.lda + pslot * 8;
            }
        }
        a_kp = p.a + ld_kt * BK;
        if (++ld_kt == nk) { ld_kt = 0; ++ld_tile; }
    };
#define A_PIECE(SET, j) ldr[SET][j] = *(const f16x8*)(a_kp + a_off[j])
    auto write_a = [&](auto set_tag, char* stage) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = prow + 32 * j;
            *(f16x8*)(stage + r * 128 + ((pslot ^ swz4(r)) << 4)) = ldr[SET][j];
        }
    };
    // ---- W loader: LDS-DMA (the weight slice of a tile column is L2-resident), one K-tile ahead.  Wave w stages the 1 KB pieces q = w + 4 j, j = 0..7:
    // staged rows i = 8 q + l / 8 = 32 j + (8 w + l / 8), physical slot l & 7 holds logical slot (l & 7) ^ swz4(i), swz4(i) = (l / 8) ^ ((4 j + w) & 7).
    // W rows are staged permuted so that a lane's 16 accumulators of a 32 x 32 block are consecutive output columns (ACT 0) or the value AND gate of 8
    // consecutive hidden units (GEGLU): staged row 8a + 4h + e  <-  source row 16h + 4a + e  |  16 (a >> 1) + 8h + 4 (a & 1) + e
    const int wi = 8 * wave + (lane >> 3);                                 // staged row inside a 32-row group
    const int wa = wi >> 3, wh = (wi >> 2) & 1, we = wi & 3;
    const int wsrc = ACT == 2 ? 16 * (wa >> 1) + 8 * wh + 4 * (wa & 1) + we : 16 * wh + 4 * wa + we;
    const int wslot[2] = {((lane & 7) ^ (lane >> 3) ^ (wave & 7)) * 8, ((lane & 7) ^ (lane >> 3) ^ ((4 + wave) & 7)) * 8};     // element offset of the logical slot, j even / odd
    const f16* w_base = p.w;
    int wd_tile = 0, wd_kt = 0;
    auto issue_w = [&](char* stage) __attribute__((always_inline)) {
        if (wd_kt == 0) {
            int tm, tn;
            tile_coords4(seq0 + wd_tile * G, p.tiles_m, p.tiles_n, p.band, tm, tn);
            w_base = p.w + (size_t)(tn * BN + wsrc) * p.K;
        }
        const f16* wp = w_base + wd_kt * BK;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(wp + (size_t)(32 * j) * p.K + wslot[j & 1], stage + 32768 + (wave + 4 * j) * 1024);
        if (++wd_kt == nk) { wd_kt = 0; ++wd_tile; }
    };
    const f16* w_kp = p.w;
    auto w_prep = [&]() __attribute__((always_inline)) {
        if (wd_kt == 0) {
            int tm, tn;
            tile_coords4(seq0 + wd_tile * G, p.tiles_m, p.tiles_n, p.band, tm, tn);
            w_base = p.w + (size_t)(tn * BN + wsrc) * p.K;
        }
        w_kp = w_base + wd_kt * BK;
        if (++wd_kt == nk) { wd_kt = 0; ++wd_tile; }
    };
#define W_PIECE(stage, j) glds16(w_kp + (size_t)(32 * (j)) * p.K + wslot[(j) & 1], (stage) + 32768 + (wave + 4 * (j)) * 1024)

    // ---- fragment read offsets: block b (0..3) of the wave's 128 rows, k16 step ks: row = w*128 + b*32 + lq, slot = 2 ks + lh
    int rd_a[4], rd_w[4];                                                  // per block: row byte offset; the slot term is added per k-step
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        rd_a[b] = (wm * 128 + b * 32 + lq) * 128;
        rd_w[b] = 32768 + (wn * 128 + b * 32 + lq) * 128;
    }
    int sz[4];                                                             // swizzle of row w*128 + b*32 + lq (the same for the A and the W region)
#pragma unroll
    for (int b = 0; b < 4; ++b) sz[b] = swz4(b * 32 + lq);

    auto read_block = [&](const int k, float (&c)[16]) __attribute__((always_inline)) {
        switch (k) {
            case 0: G4_READ_0(c); break; case 1: G4_READ_1(c); break; case 2: G4_READ_2(c); break; case 3: G4_READ_3(c); break;
            case 4: G4_READ_4(c); break; case 5: G4_READ_5(c); break; case 6: G4_READ_6(c); break; case 7: G4_READ_7(c); break;
            case 8: G4_READ_8(c); break; case 9: G4_READ_9(c); break; case 10: G4_READ_10(c); break; case 11: G4_READ_11(c); break;
            case 12: G4_READ_12(c); break; case 13: G4_READ_13(c); break; case 14: G4_READ_14(c); break; default: G4_READ_15(c); break;
        }
    };

    // ---- prologue: K-tiles 0 and 1 requested, tile 0 stored to stage 0
    issue_w(smem);
    issue_loads(std::integral_constant<int, 0>{});
    if (V > 1) { issue_loads(std::integral_constant<int, 1>{}); issue_w(smem + STAGE); }
    if (V > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // K-tile 0 (DMA + A pieces) has landed
    write_a(std::integral_constant<int, 0>{}, smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // fragments: two sets, the reads of k-step s + 1 are issued before the MFMAs of k-step s
    f16x8 fa[2][4], fw[2][4];
    auto read_frags = [&](const char* stage, const int ks, auto set_tag) __attribute__((always_inline)) {
        constexpr int FS = decltype(set_tag)::value;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            fa[FS][b] = *(const f16x8*)(stage + rd_a[b] + (((2 * ks + lh) ^ sz[b]) << 4));
            fw[FS][b] = *(const f16x8*)(stage + rd_w[b] + (((2 * ks + lh) ^ sz[b]) << 4));
        }
    };
    read_frags(smem, 0, std::integral_constant<int, 0>{});

    int cur_tile = 0, cur_kt = 0;
    auto step = [&](const int v, auto par_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;                      // v & 1: LDS stage of K-tile v; register set PAR holds K-tile v + 2 after this step's loads
        const char* cur = smem + PAR * STAGE;
        char* nxt = smem + (PAR ^ 1) * STAGE;
        auto wr = [&](const int j) __attribute__((always_inline)) {        // A piece j of K-tile v + 1 (register set PAR ^ 1) -> the other stage
            const int r = prow + 32 * j;
            *(f16x8*)(nxt + r * 128 + ((pslot ^ swz4(r)) << 4)) = ldr[PAR ^ 1][j];
        };
        const bool more = v + 1 < V, more2 = v + 2 < V;
        // One memory instruction between every two MFMAs (an in-order wave that issues a burst of other instructions leaves the matrix pipe idle:
        // MI355X_MICROARCH, "<= 5 fillers per 32-cycle MFMA gap").  Per K-tile: 64 MFMAs, 32 fragment reads, 8 DMA pieces, 8 global loads, 8 LDS stores.
#define RDA(FS, st, ks, b) fa[FS][b] = *(const f16x8*)((st) + rd_a[b] + (((2 * (ks) + lh) ^ sz[b]) << 4))
#define RDW(FS, st, ks, b) fw[FS][b] = *(const f16x8*)((st) + rd_w[b] + (((2 * (ks) + lh) ^ sz[b]) << 4))
#define MM(MF, k, i, j, FS) MF##_##k(fw[FS][i], fa[FS][j])
        // ---- k-step 0 (fragment set 0): fragments of k-step 1 -> set 1; W of K-tile v + 1 by DMA into the other stage (free since the barrier of step v - 1)
        // One memory instruction behind every MFMA.  Stream order (vmcnt): W(v+1) [k-step 3 of step v-1], A(v+2) [k-step 0], barrier of step v with vmcnt(8).
        // ---- k-step 0 (set 0): fragments of k-step 1 -> set 1; A of K-tile v + 2 into register set PAR (it held K-tile v: in LDS since step v - 1)
        if (more2) a_prep();                    // past the last K-tile the pieces re-fetch the previous addresses into a dead register set / the free stage: no branches in the MFMA stream
#define GRP0(MF)                                                                                                                                  \
        MM(MF, 0, 0, 0, 0); RDA(1, cur, 1, 0); MM(MF, 1, 0, 1, 0); A_PIECE(PAR, 0); MM(MF, 2, 0, 2, 0); RDA(1, cur, 1, 1); MM(MF, 3, 0, 3, 0); A_PIECE(PAR, 1); \
        MM(MF, 4, 1, 0, 0); RDA(1, cur, 1, 2); MM(MF, 5, 1, 1, 0); A_PIECE(PAR, 2); MM(MF, 6, 1, 2, 0); RDA(1, cur, 1, 3); MM(MF, 7, 1, 3, 0); A_PIECE(PAR, 3); \
        MM(MF, 8, 2, 0, 0); RDW(1, cur, 1, 0); MM(MF, 9, 2, 1, 0); A_PIECE(PAR, 4); MM(MF, 10, 2, 2, 0); RDW(1, cur, 1, 1); MM(MF, 11, 2, 3, 0); A_PIECE(PAR, 5); \
        MM(MF, 12, 3, 0, 0); RDW(1, cur, 1, 2); MM(MF, 13, 3, 1, 0); A_PIECE(PAR, 6); MM(MF, 14, 3, 2, 0); RDW(1, cur, 1, 3); MM(MF, 15, 3, 3, 0); A_PIECE(PAR, 7)
        if (cur_kt == 0) { GRP0(G4_MFMA0); } else { GRP0(G4_MFMA); }        // first k-step of an output tile: C = 0
        // ---- k-step 1 (set 1): fragments of k-step 2 -> set 0
        MM(G4_MFMA, 0, 0, 0, 1); RDA(0, cur, 2, 0); MM(G4_MFMA, 1, 0, 1, 1); MM(G4_MFMA, 2, 0, 2, 1); RDA(0, cur, 2, 1); MM(G4_MFMA, 3, 0, 3, 1);
        MM(G4_MFMA, 4, 1, 0, 1); RDA(0, cur, 2, 2); MM(G4_MFMA, 5, 1, 1, 1); MM(G4_MFMA, 6, 1, 2, 1); RDA(0, cur, 2, 3); MM(G4_MFMA, 7, 1, 3, 1);
        MM(G4_MFMA, 8, 2, 0, 1); RDW(0, cur, 2, 0); MM(G4_MFMA, 9, 2, 1, 1); MM(G4_MFMA, 10, 2, 2, 1); RDW(0, cur, 2, 1); MM(G4_MFMA, 11, 2, 3, 1);
        MM(G4_MFMA, 12, 3, 0, 1); RDW(0, cur, 2, 2); MM(G4_MFMA, 13, 3, 1, 1); MM(G4_MFMA, 14, 3, 2, 1); RDW(0, cur, 2, 3); MM(G4_MFMA, 15, 3, 3, 1);
        // ---- k-step 2 (set 0): fragments of k-step 3 -> set 1; the 8 A pieces of K-tile v + 1 (register set PAR ^ 1) into the other stage
        MM(G4_MFMA, 0, 0, 0, 0); RDA(1, cur, 3, 0); MM(G4_MFMA, 1, 0, 1, 0); wr(0); MM(G4_MFMA, 2, 0, 2, 0); RDA(1, cur, 3, 1); MM(G4_MFMA, 3, 0, 3, 0); wr(1);
        MM(G4_MFMA, 4, 1, 0, 0); RDA(1, cur, 3, 2); MM(G4_MFMA, 5, 1, 1, 0); wr(2); MM(G4_MFMA, 6, 1, 2, 0); RDA(1, cur, 3, 3); MM(G4_MFMA, 7, 1, 3, 0); wr(3);
        MM(G4_MFMA, 8, 2, 0, 0); RDW(1, cur, 3, 0); MM(G4_MFMA, 9, 2, 1, 0); wr(4); MM(G4_MFMA, 10, 2, 2, 0); RDW(1, cur, 3, 1); MM(G4_MFMA, 11, 2, 3, 0); wr(5);
        MM(G4_MFMA, 12, 3, 0, 0); RDW(1, cur, 3, 2); MM(G4_MFMA, 13, 3, 1, 0); wr(6); MM(G4_MFMA, 14, 3, 2, 0); RDW(1, cur, 3, 3); MM(G4_MFMA, 15, 3, 3, 0); wr(7);
        // every fragment read of K-tile v has been issued and K-tile v + 1 is stored: publish it, free this stage.  A bare s_barrier: __syncthreads()
        // carries a release fence that the compiler lowers to vmcnt(0) because of the LDS-DMA, which would drain the 8 A loads every K-tile.
        // The W DMA of K-tile v + 1 has landed with vmcnt(8): it is older than this step's 8 A loads, which stay in flight.
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- k-step 3 (set 1), with the first fragments of K-tile v + 1 read underneath -> set 0, and W of K-tile v + 2 by DMA into THIS stage (just freed)
        if (more2) w_prep();
        MM(G4_MFMA, 0, 0, 0, 1); RDA(0, nxt, 0, 0); MM(G4_MFMA, 1, 0, 1, 1); W_PIECE((char*)cur, 0); MM(G4_MFMA, 2, 0, 2, 1); RDA(0, nxt, 0, 1); MM(G4_MFMA, 3, 0, 3, 1); W_PIECE((char*)cur, 1);
        MM(G4_MFMA, 4, 1, 0, 1); RDA(0, nxt, 0, 2); MM(G4_MFMA, 5, 1, 1, 1); W_PIECE((char*)cur, 2); MM(G4_MFMA, 6, 1, 2, 1); RDA(0, nxt, 0, 3); MM(G4_MFMA, 7, 1, 3, 1); W_PIECE((char*)cur, 3);
        MM(G4_MFMA, 8, 2, 0, 1); RDW(0, nxt, 0, 0); MM(G4_MFMA, 9, 2, 1, 1); W_PIECE((char*)cur, 4); MM(G4_MFMA, 10, 2, 2, 1); RDW(0, nxt, 0, 1); MM(G4_MFMA, 11, 2, 3, 1); W_PIECE((char*)cur, 5);
        MM(G4_MFMA, 12, 3, 0, 1); RDW(0, nxt, 0, 2); MM(G4_MFMA, 13, 3, 1, 1); W_PIECE((char*)cur, 6); MM(G4_MFMA, 14, 3, 2, 1); RDW(0, nxt, 0, 3); MM(G4_MFMA, 15, 3, 3, 1); W_PIECE((char*)cur, 7);
        if (++cur_kt == nk) {
            // ---------------- epilogue of output tile cur_tile ----------------
            int tm, tn;
            tile_coords4(seq0 + cur_tile * G, p.tiles_m, p.tiles_n, p.band, tm, tn);
            const int n0 = tn * BN + wn * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // bias of this lane's 16 rows of block i, in the order of the accumulator registers: two 16-byte loads
                float bv[16];
                {
                    f16x8 b0 = {0, 0, 0, 0, 0, 0, 0, 0}, b1 = b0;
                    if (p.bias) {
                        const f16* bp = p.bias + n0 + i * 32 + (ACT == 2 ? 8 * lh : 16 * lh);
                        b0 = *(const f16x8*)bp;                            // ACT 0: columns 16 lh + 0..7  | GEGLU: value rows 8 lh + 0..7
                        b1 = *(const f16x8*)(bp + (ACT == 2 ? 16 : 8));   //        columns 16 lh + 8..15 |        gate rows 16 + 8 lh + 0..7
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) { bv[r] = (float)b0[r]; bv[r + 8] = (float)b1[r]; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = tm * BM + wm * 128 + j * 32 + lq;
                    float c[16];
                    read_block(4 * i + j, c);
                    if constexpr (ACT == 2) {
                        f16x8 o;
#pragma unroll
                        for (int r = 0; r < 8; r += 2) {
                            const f32x2 val = {c[r] + bv[r], c[r + 1] + bv[r + 1]}, gate = {c[r + 8] + bv[r + 8], c[r + 9] + bv[r + 9]};
                            const f32x2 y = ew_vgelu2(val, gate) * p.c_acc;
                            o[r] = (f16)y[0];
                            o[r + 1] = (f16)y[1];
                        }
                        if (m < p.M) *(f16x8*)(p.out + (size_t)m * p.ld_out + ((n0 + i * 32) >> 1) + 8 * lh) = o;
                    } else {
                        f16x8 o0, o1;
#pragma unroll
                        for (int r = 0; r < 8; ++r) { o0[r] = (f16)(p.c_acc * (c[r] + bv[r])); o1[r] = (f16)(p.c_acc * (c[r + 8] + bv[r + 8])); }
                        if (m < p.M) {
                            f16* op = p.out + (size_t)m * p.ld_out + n0 + i * 32 + 16 * lh;
                            *(f16x8*)op = o0;
                            *(f16x8*)(op + 8) = o1;
                        }
                    }
                }
            }
            cur_kt = 0;
            ++cur_tile;
        }
    };
    int v = 0;
    for (; v + 1 < V; v += 2) {
        step(v, std::integral_constant<int, 0>{});
        step(v + 1, std::integral_constant<int, 1>{});
    }
    if (v < V) step(v, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the unconditional prefetches of the last step
}

}  // namespace

extern char g_gemm_last_kernel[64];

bool ew_gemm4_wants(const GemmP& p) {
    return p.mode == EW_A_DENSE && p.c2 == 0 && p.N % BN == 0 && p.K % BK == 0 && p.K >= 2 * BK && !p.rowbias && !p.r1 && !p.r2 && !p.out_lo &&
           (p.act == EW_ACT_NONE || p.act == EW_ACT_GEGLU) && p.M >= 2048 && (long long)p.M * p.lda < (1LL << 31);
}

ew_status ew_gemm4_dispatch(const GemmP& p, hipStream_t s) {
    GemmP q = p;
    q.tiles_m = ew_cdiv(p.M, BM);
    q.tiles_n = p.N / BN;
    q.band = ((long long)p.N * p.K * 2 > 3LL * 1024 * 1024) ? 4 : 0;
    const long long tiles = (long long)q.tiles_m * q.tiles_n;
    int grid = ew_cu_budget();
    if (tiles < grid) grid = (int)((tiles + 7) / 8 * 8);
    const int lds = 2 * STAGE;
    snprintf(g_gemm_last_kernel, 64, "gemm4_kernel<%d>", p.act == EW_ACT_GEGLU ? 2 : 0);
    if (p.act == EW_ACT_GEGLU) {
        static std::atomic<unsigned long long> mask{0};
        if (ew_status st = ew_ensure_dynamic_lds((const void*)gemm4_kernel<2>, lds, mask)) return st;
        hipLaunchKernelGGL(gemm4_kernel<2>, dim3(grid), dim3(64 * NW), lds, s, q);
    } else {
        static std::atomic<unsigned long long> mask{0};
        if (ew_status st = ew_ensure_dynamic_lds((const void*)gemm4_kernel<0>, lds, mask)) return st;
        hipLaunchKernelGGL(gemm4_kernel<0>, dim3(grid), dim3(64 * NW), lds, s, q);
    }
    return ew_check_launch("ew_gemm_f16(gen4)");
}
