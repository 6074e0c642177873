// ff_fused.hip -- fused GEGLU feed-forward for the 320-channel (level 0) transformer blocks, gfx950 (round 3).
//
//   out = c_acc * ( GEGLU(x W1^T + b1) W2^T + b2 + rowbias ) + c_r1 * r1 + c_r2 * r2,   x = the LayerNorm output (ew_layernorm_f16)
//   (round 6: the LayerNorm-in-the-prologue variants -- ln_gamma prologue, +9 ms per forward; folded LayerNorm, +1.1 ms -- were removed from the
//   source; they live in git history, commit 3205fa6, and their measurements in DESIGN.md sections 3.3-3.5)
//
// i.e. diffusers FeedForward(dim, activation_fn="geglu") = net.0 (GEGLU proj 320 -> 2 x 1280) + net.2 (Linear 1280 -> 320) of
// BasicTransformerBlock.ff / TemporalBasicTransformerBlock.ff / .ff_in with the LayerNorm in front of it (norm3 / norm_in;
// instantiated through evoworld/trainer/unet_plucker.py:161-233) and the residual / AlphaBlender
// epilogue of ew_gemm_f16.
//
// Why a fused kernel: as LayerNorm + two GEMMs the 1280-wide GEGLU intermediate of level 0 (460800 x 1280 fp16 = 1.18 GB) is
// written and read back 15 times per forward and the normalised tensor 15 times more.  Here neither leaves the chip:
//   * one workgroup = 8 waves as 4 (M) x 2 (N), two per SIMD, on a 128-row tile.  A wave owns 32 rows; their (normalised) x
//     fragments stay in 80 registers for the whole tile.  Register
//     budget as compiled (-Rpass-analysis=kernel-resource-usage, round 4): 256 VGPRs, 0 AGPRs, 17-55 spilled VGPRs
//     (72-224 B of scratch per lane) --
//     every scratch access sits in the per-TILE prologue / epilogue; the 40-iteration chunk loop itself holds x fragments, both
//     accumulator sets and the W fragments in registers with zero scratch traffic (checked in the ISA: no scratch_load /
//     scratch_store between the loop header and its back edge).  (Until round 4 the ln_gamma prologue was a run-time branch
//     inside every variant: dead code in the shipped configuration, but it cost the plain variants ~170 more spilled VGPRs.);
//   * the hidden dimension is walked in 40 chunks of 32.  Up-projection of a chunk: the wave computes the value + gate columns
//     of ITS half of the chunk (16 hidden units: 2 fragments x 2 row fragments x 10 k-steps = 40 MFMAs), applies bias and
//     value * gelu(gate) in registers and leaves its four halves per lane in a small LDS exchange buffer; after the chunk's
//     barrier either wave of the pair reads the complete A fragment (32 hidden units) back with one ds_read_b128 per row fragment
//     and multiplies it into ITS half of the 320 output columns (10 fragments x 2 = 20 MFMAs, accumulator 32 x 160 = 80
//     registers).  W1's rows are packed so that the four outputs a lane owns per (value, gate) fragment pair are consecutive
//     hidden indices -- the GEGLU result is already in A-fragment order;
//   * the down-projection trails the up-projection by one chunk, so ONE barrier per chunk serves both the weight hand-over
//     and the GEGLU exchange; the GEGLU arithmetic of chunk c-1 (VALU: two transcendentals per element) is issued between the
//     up-projection MFMAs of chunk c (1), the DMA requests between the down-projection MFMAs;
//   * weights go through LDS one chunk image (W1 40 KB + W2 20 KB) per step, double-buffered, by LDS-DMA from host-packed
//     images that are byte copies of the LDS layout (1 KB contiguous per DMA instruction; the XOR swizzles that make the
//     ds_read_b128 fragment reads conflict-free are baked into the pack).  A chunk's image is requested a whole chunk before it is
//     needed.  The 2.4 MB of weights are shared by all 256 workgroups walking them in step: L2-resident;
//   * the epilogue is the LDS-free "direct" form of gemm3_f16.hip (W2's rows are staged permuted so that a lane's accumulator
//     pair is 8 consecutive output columns); all residual operands of a row fragment are requested before the first is used.
// Same rounding points as the separate kernels (LayerNorm output rounded to fp16, fp32 accumulation, GEGLU in fp32, intermediate
// rounded to fp16 once, fp32 accumulation, output split or fp16): results agree to fp32 summation order.
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int C = 320, HID = 1280, CH = 32, NCH = HID / CH;       // 40 chunks of 32 hidden units
constexpr int FF_WM = 4;  // waves along M: 4 -> 8 waves (2 per SIMD, 256 registers), 32 rows each (2 -> 4 waves, 1 per SIMD, 64 rows each measured 1.53 vs 1.36 ms)
constexpr int BM = 128, WAVES_N = 2, NWV = FF_WM * WAVES_N, WROWS = BM / FF_WM, RF = WROWS / 16;
constexpr int KS = C / 32;                                         // 10 k-steps of the up-projection
constexpr int KT = C / 64;                                         // W1 chunk image = 5 K-tiles of [64 staged rows][64 k]
constexpr int W1_KT = 64 * 128;                                    // 8 KB
constexpr int W1_TILE = KT * W1_KT;                                // 40 KB per chunk
constexpr int W2_TILE = C * CH * 2;                                // 320 staged rows x 32 k x 2 B = 20 KB per chunk
constexpr int HX_TILE = (BM / 16) * 64 * 16;                       // GEGLU exchange: 8 row fragments x 64 lanes x 16 B = 8 KB per chunk
constexpr int LDS_W1 = 0, LDS_W2 = 2 * W1_TILE, LDS_HX = LDS_W2 + 2 * W2_TILE, LDS_B1 = LDS_HX + 2 * HX_TILE;
constexpr int LDS_SCR = LDS_B1 + 2 * HID * 4;             // b1 as fp32 (enters through the C operand); then a 4 KB scratch slot (GEGLU output of the no-op first slice)
constexpr int LDS_BYTES = LDS_SCR + 64 * NWV * 8 + 4096;                    // 80 + 40 + 16 + 5 KB
constexpr int P1 = W1_TILE / 1024 / NWV;                           // W1 DMA pieces per wave and chunk (5); W2's 20 pieces are dealt round-robin
constexpr int P2MAX = (W2_TILE / 1024 + NWV - 1) / NWV;            // 3 (waves 4-7 issue 2)
constexpr int NJ2 = C / 16 / WAVES_N;                              // 10 output fragments per wave

struct FfP {
    const f16* x;
    const f16* w1p;
    const f16* b1p;
    const f16* w2p;
    const f16* b2;
    const f16* rowbias;
    const f16* r1;
    const f16* r2;
    const int8_t* r1_lo;
    const int8_t* r2_lo;
    f16* out;
    int8_t* out_lo;
    const f16* zero_page;
    int M, rows_per_group, ld_rowbias, n_tiles;
    float c_acc, c_r1, c_r2;
};

#define FF_FENCE() asm volatile("" ::: "memory")
// vmcnt(0) through the builtin (simm16: vmcnt = 0, expcnt = 7, lgkmcnt = 15), not inline asm: hipcc's waitcnt pass then KNOWS the
// queue is empty.  With an opaque asm wait it kept a vmcnt(0) in front of the first use of the x fragments in EVERY chunk (their
// loads are issued at the end of the previous tile), which drained the weight DMA right after it was requested: 3.4x slower.
#define FF_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)
#define FF_PIN() __builtin_amdgcn_sched_barrier(0)
#define FF_WAIT_VM0_LGKM0() __builtin_amdgcn_s_waitcnt(0x0070)

template <bool LO, bool R2>
__global__ __launch_bounds__(64 * NWV, NWV / 4) void ff320_kernel(const FfP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)(tid & 63) * 16u;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fks = lane >> 4, sw = frow & 7;
    if (wave >= NWV / 2) __builtin_amdgcn_s_setprio(1);

    // W2 and exchange buffers start as zeros: the first chunk of a tile runs the trailing down-projection slot on a zero A
    // fragment (no branch in the pipeline), which must not meet NaN bit patterns of uninitialised LDS
    for (int i = tid; i < (2 * W2_TILE + 2 * HX_TILE) / 16; i += 64 * NWV) *(f16x8*)(smem + LDS_W2 + i * 16) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    // packed b1 -> LDS once (5 KB)
    for (int i = tid; i < 2 * HID / 8; i += 64 * NWV) {
        const f16x8 b = *(const f16x8*)((const char*)p.b1p + i * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) ((float*)(smem + LDS_B1))[i * 8 + e] = (float)b[e];
    }

    const int G = gridDim.x;
    const int n_my = (p.n_tiles - (int)blockIdx.x + G - 1) / G;            // tiles blockIdx.x, +G, ...
    if (n_my <= 0) return;
    const int CC_total = n_my * NCH;
    // Start stagger.  All tiles cost the same, so 256 workgroups started together stay in step for the whole launch: every CU asks
    // L2 for the same 60 KB weight chunk at the same moment and every CU's epilogue (the only HBM traffic) falls into the same few
    // microseconds.  n_tiles is rarely a multiple of the grid: workgroups that own one tile fewer than the busiest have a whole tile
    // time of slack, and spend part of it up front -- (blockIdx mod 16) x ~4 us (s_sleep 127 = 8128 clocks)
    if (n_my < (p.n_tiles + G - 1) / G) {
        const int d = (int)(blockIdx.x % 16);
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ---- weight DMA stream: global chunk cc <-> W1 / W2 chunk images (cc mod 40), identical for every tile; buffers cc & 1
    auto issue_w1 = [&](int c_mod, int slot) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t r1d = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w1p + (size_t)c_mod * W1_TILE), 0, W1_TILE, 0x00020000);
        char* d1 = smem + LDS_W1 + slot * W1_TILE + wave * (P1 * 1024);
#pragma unroll
        for (int k = 0; k < P1; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1d, (lptr_t)(d1 + k * 1024), 16, (int)lane16, wave * (P1 * 1024) + k * 1024, 0, 0);
    };
    issue_w1(0, 0);
    int cc = 0;                                 // global chunk index
    FF_WAIT_VM0_LGKM0();                        // every wave drains ITS OWN DMA pieces of W1(0) (vmcnt) and its LDS writes before the
    __syncthreads();                            // barrier: the fragment reads below cover pieces other waves loaded (b1 staged too)
    if (CC_total > 1) issue_w1(1, 1);

    // fragment read offsets.  W1 K-tile image: staged row R = j*16 + frow (128-byte rows), 16-byte k-slot ks in 0..7 ->
    // R*128 + ((ks ^ (R & 7)) << 4); W2 chunk image: staged row R = jj*16 + frow (64-byte rows), k-slot ks in 0..3 ->
    // R*64 + ((ks ^ g((R >> 2) & 3)) << 4), g = {0, 2, 3, 1}: four rows share a 256-byte bank row, and the 16 lanes a
    // ds_read_b128 services together are (fks, rows 0-3 | 12-15) + (fks+1, rows 4-11) -- g makes their 16 slots distinct
    int rd1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) rd1[h] = (2 * wn * 16 + frow) * 128 + (((h * 4 + fks) ^ sw) << 4);      // this wave's fragments j = 2 wn, 2 wn + 1
    const int gq = (0x78 >> (2 * ((frow >> 2) & 3))) & 3;             // {0, 2, 3, 1}[(frow >> 2) & 3]
    const int rd2 = (wn * NJ2 * 16 + frow) * 64 + ((fks ^ gq) << 4);                                    // this wave's fragments jj = 10 wn ...
    // exchange buffer: row fragment (wm, rf), lane L: 16 bytes = the A fragment of the chunk's k-step; this wave writes bytes 8 wn ..
    const int hx_off = (wm * RF * 64 + lane) * 16;
#define FF_W1F(buf, ks_, j_) (*(const f16x8*)((buf) + ((ks_) >> 1) * W1_KT + (j_) * 2048 + rd1[(ks_) & 1]))

    // W1 fragment ring: k-step ks uses wf[ks % 3]; the fragments of ks+2 are read while ks is multiplied; the first two sets of
    // a chunk are read during the second phase of the PREVIOUS chunk.
    f16x8 wf[3][2];
    {
        const char* w1b = smem + LDS_W1;
#pragma unroll
        for (int j = 0; j < 2; ++j) { wf[0][j] = FF_W1F(w1b, 0, j); wf[1][j] = FF_W1F(w1b, 1, j); }
    }

    // ---- x fragments of this wave's 32 rows of a tile: row frow (+16 rf), k = ks*32 + fks*8 .. +8 (both waves of an M pair hold
    // the same rows).  The NEXT tile's fragments are requested before the epilogue of the current one.
    f16x8 xf[RF][KS];
    auto load_x = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) {
            const int m = min(tile * BM + wm * WROWS + rf * 16 + frow, p.M - 1);
            const f16* xp = p.x + (size_t)m * C + fks * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[rf][ks] = *(const f16x8*)(xp + ks * 32);
        }
    };
    load_x((int)blockIdx.x);
    FF_WAIT_VM0();

    for (int ti = 0; ti < n_my; ++ti) {
        const int tile = (int)blockIdx.x + ti * G;
        const int m_w0 = tile * BM + wm * WROWS;
        f32x4 acc2[RF][NJ2];
#pragma unroll
        for (int rf = 0; rf < RF; ++rf)
#pragma unroll
            for (int jj = 0; jj < NJ2; ++jj) acc2[rf][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f16x8 hf_old[RF] = {};                  // complete GEGLU A fragments of the previous chunk (zero before the tile's first)
        f32x4 acc1p[RF][2] = {};                // 1: up-projection accumulators of the previous chunk, waiting for their GEGLU

        // Software pipeline over the chunks of the tile.  Iteration c:
        //   phase 1: up-projection of chunk c, this wave's half (40 MFMAs; W1 fragments two k-steps ahead), with the GEGLU of chunk c-1
        //            between the MFMAs -> exchange buffer (cc - 1) & 1
        //   barrier: W1 buffer cc & 1 and W2 buffer cc & 1 are free (down-projection of chunk c-2 is done); this wave's pieces of
        //            W1(cc+1) and W2(cc-1), requested a chunk ago, have landed; the exchange halves of chunk c-1 are complete
        //   phase 2: read the A fragments of chunk c-1 from the exchange buffer; down-projection of chunk c-1 (20 MFMAs); DMA requests
        //            for W1(cc+2) / W2(cc)
        // The down-projection trails by one chunk; it is flushed (with the last chunk's GEGLU) after the last chunk of the tile.
        // Segment trace (tools/experiments/exp39_ff_trace.py, -DFF_TRACE): a chunk is ~4300 clocks per wave against 1920 of MFMA pipe
        // time per SIMD; the two waves of a SIMD share its VALU issue (2 x ~700 clocks of GEGLU) and the older one wins the arbitration,
        // so the younger's phases run late and the older idles ~1100 clocks at each barrier.  A forced half-chunk offset between the two
        // halves (ping-pong, two barriers per chunk) measured 12 % slower, scalar instead of packed fp32 GEGLU 3 % slower.
        // bias + GEGLU of one row fragment of this wave's half chunk: fragments (value 2 wn, gate 2 wn + 1) -> hidden 8 fks + 4 wn + e
        auto geglu_rf = [&](const f32x4 (&a)[2], const char* bb, char* dst) __attribute__((always_inline)) {
            const f32x4 va = a[0], gg = a[1];          // the bias came in through the accumulators' initial value
            const f32x2 o01 = ew_vgelu2((f32x2){va[0], va[1]}, (f32x2){gg[0], gg[1]});
            const f32x2 o23 = ew_vgelu2((f32x2){va[2], va[3]}, (f32x2){gg[2], gg[3]});
            const f16x4 o4 = {(f16)o01[0], (f16)o01[1], (f16)o23[0], (f16)o23[1]};
            *(f16x4*)dst = o4;
        };
        for (int c = 0; c < NCH; ++c, ++cc) {
            const char* w1b = smem + LDS_W1 + (cc & 1) * W1_TILE;
            const char* w1n = smem + LDS_W1 + ((cc + 1) & 1) * W1_TILE;
            const char* w2b = smem + LDS_W2 + ((cc + 1) & 1) * W2_TILE;       // W2(cc - 1)
            f32x4 acc1[RF][2];
            {   // this chunk's bias (value fragment, gate fragment: 4 consecutive hidden units of this lane each) = the C operand
                const char* bc = smem + LDS_B1 + (c * 64 + 2 * wn * 16 + fks * 4) * 4;
                const f32x4 b0 = *(const f32x4*)bc, b1v = *(const f32x4*)(bc + 64);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) { acc1[rf][0] = b0; acc1[rf][1] = b1v; }
            }
            // 1: the GEGLU of the PREVIOUS chunk (VALU, ~90 instructions with two transcendentals per element) is issued between
            // the up-projection MFMAs of this one, one row fragment per half of the k loop; its halves go to exchange buffer (cc - 1) & 1 and are
            // complete at this chunk's barrier.  On the first chunk of a tile the accumulators are zeros and the result goes to a scratch slot.
            [[maybe_unused]] const char* bbp = smem + LDS_B1 + ((c > 0 ? c - 1 : 0) * 64 + 2 * wn * 16 + fks * 4) * 2;
            [[maybe_unused]] char* hxp = c > 0 ? smem + LDS_HX + ((cc + 1) & 1) * HX_TILE + hx_off + wn * 8 : smem + LDS_SCR + tid * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 2 < KS) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) wf[(ks + 2) % 3][j] = FF_W1F(w1b, ks + 2, j);
                }
                FF_PIN();
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf)
                        acc1[rf][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks % 3][j], xf[rf][ks], acc1[rf][j], 0, 0, 0);
                static_assert(RF == 2 || RF == 4, "GEGLU slices");
                if constexpr (RF == 2) {
                    if (ks == 1) geglu_rf(acc1p[0], bbp, hxp);
                    if (ks == 5) geglu_rf(acc1p[1], bbp, hxp + 1024);
                } else {
                    if (ks == 1 || ks == 3 || ks == 5 || ks == 7) geglu_rf(acc1p[(ks - 1) / 2], bbp, hxp + ((ks - 1) / 2) * 1024);
                }
                FF_PIN();
            }
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) { acc1p[rf][0] = acc1[rf][0]; acc1p[rf][1] = acc1[rf][1]; }
            FF_WAIT_VM0_LGKM0();
            FF_FENCE();
            __builtin_amdgcn_s_barrier();
            FF_FENCE();
            // ---- phase 2 (no branches: on the first chunk of a tile the exchange buffer holds zeros -- see the flush -- and past the
            // end of the block's work the W1 request / fragment reads touch buffers nobody reads again)
            if (c > 0) {
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) hf_old[rf] = *(const f16x8*)(smem + LDS_HX + ((cc + 1) & 1) * HX_TILE + hx_off + rf * 1024);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) { wf[0][j] = FF_W1F(w1n, 0, j); wf[1][j] = FF_W1F(w1n, 1, j); }
            // DMA requests of this hand-over: W1(cc+2) -> buffer cc & 1 (5 pieces per wave), W2(cc) -> buffer cc & 1 (pieces wave + 8k)
            int c2 = c + 2;
            c2 = c2 >= NCH ? c2 - NCH : c2;
            // buffer form (round 6): descriptor = the chunk image, lane offset constant, piece offset scalar -- no 64-bit pointer adds on the VALU
            const __amdgpu_buffer_rsrc_t r1d = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w1p + (size_t)c2 * W1_TILE), 0, W1_TILE, 0x00020000);
            char* d1 = smem + LDS_W1 + (cc & 1) * W1_TILE + wave * (P1 * 1024);
            const __amdgpu_buffer_rsrc_t r2d = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w2p + (size_t)c * W2_TILE), 0, W2_TILE, 0x00020000);
            char* d2 = smem + LDS_W2 + (cc & 1) * W2_TILE + wave * 1024;
            FF_PIN();
            // ---- down-projection of chunk c-1: one k-step x this wave's 10 output fragments x 2 row fragments
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                f16x8 w2f[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) w2f[u] = *(const f16x8*)(w2b + (g2 * 5 + u) * 1024 + rd2);
                FF_PIN();
#pragma unroll
                for (int u = 0; u < 5; ++u)
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf)
                        acc2[rf][g2 * 5 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[u], hf_old[rf], acc2[rf][g2 * 5 + u], 0, 0, 0);
                FF_PIN();
                // DMA pieces of this wave, half per group
                constexpr int PER = (P1 + P2MAX + 1) / 2;
#pragma unroll
                for (int k = g2 * PER; k < g2 * PER + PER && k < P1 + P2MAX; ++k) {
                    if (k < P1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1d, (lptr_t)(d1 + k * 1024), 16, (int)lane16, wave * (P1 * 1024) + k * 1024, 0, 0);
                    else if (wave + NWV * (k - P1) < W2_TILE / 1024)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2d, (lptr_t)(d2 + (k - P1) * NWV * 1024), 16, (int)lane16, wave * 1024 + (k - P1) * NWV * 1024, 0, 0);
                }
                FF_PIN();
            }
        }
        // ---- flush: down-projection of the tile's last chunk (its W2 image was requested in the last phase 2, its exchange halves
        // were written there); the exchange buffer the NEXT tile's first chunk will read as "chunk -1" is zeroed
        {
            {   // GEGLU of the tile's last chunk (nothing left to hide it under)
                const char* bb = smem + LDS_B1 + ((NCH - 1) * 64 + 2 * wn * 16 + fks * 4) * 2;
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) {
                    geglu_rf(acc1p[rf], bb, smem + LDS_HX + ((cc + 1) & 1) * HX_TILE + hx_off + rf * 1024 + wn * 8);
                    acc1p[rf][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    acc1p[rf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            FF_WAIT_VM0_LGKM0();
            FF_FENCE();
            __builtin_amdgcn_s_barrier();
            FF_FENCE();
            const char* w2b = smem + LDS_W2 + ((cc + 1) & 1) * W2_TILE;       // W2(cc - 1): cc already points past the tile
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) hf_old[rf] = *(const f16x8*)(smem + LDS_HX + ((cc + 1) & 1) * HX_TILE + hx_off + rf * 1024);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                f16x8 w2f[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) w2f[u] = *(const f16x8*)(w2b + (g2 * 5 + u) * 1024 + rd2);
#pragma unroll
                for (int u = 0; u < 5; ++u)
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf)
                        acc2[rf][g2 * 5 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[u], hf_old[rf], acc2[rf][g2 * 5 + u], 0, 0, 0);
            }
        }
        // ---- epilogue: lane (frow, fks) owns, for pair q of this wave's column half, the 8 output columns q*32 + fks*8 .. +8 of rows
        // frow, frow + 16.  All residual operands of a row fragment are requested before the first one is used.
        {
            const f16* rbp = p.rowbias ? p.rowbias : p.zero_page;
            const f16* r1p = p.r1 ? p.r1 : p.zero_page;
            const f16* r2p = p.r2 ? p.r2 : p.zero_page;
            const int8_t* r1lp = p.r1_lo ? p.r1_lo : (const int8_t*)p.zero_page;
            const int8_t* r2lp = p.r2_lo ? p.r2_lo : (const int8_t*)p.zero_page;
            const int mrb = p.rowbias ? 1 : 0, m1 = p.r1 ? 1 : 0, m2 = p.r2 ? 1 : 0, m1l = p.r1_lo ? 1 : 0, m2l = p.r2_lo ? 1 : 0;
            constexpr int NQ = NJ2 / 2;
            const int n_w0 = wn * (C / WAVES_N) + fks * 8;
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) {
                const int m = m_w0 + rf * 16 + frow;
                const int mc = min(m, p.M - 1);
                const int g = mc / p.rows_per_group;
                f16x8 q1v[NQ], q2v[NQ];
                // the next tile's x fragments are requested once half of the accumulator is dead (register budget), behind the
                // last row fragment's residual loads (vmcnt is in order: those are waited for with the x loads still in flight)
                auto next_x = [&]() __attribute__((always_inline)) { if (rf == RF - 1 && ti + 1 < n_my) load_x(tile + G); };
                u32x2 q1l[NQ], q2l[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int n = n_w0 + q * 32;
                    q1v[q] = *(const f16x8*)(r1p + ((size_t)mc * C + n) * m1);
                    if (LO) q1l[q] = *(const u32x2*)(r1lp + ((size_t)mc * C + n) * m1l);
                    if (R2) {
                        q2v[q] = *(const f16x8*)(r2p + ((size_t)mc * C + n) * m2);
                        if (LO) q2l[q] = *(const u32x2*)(r2lp + ((size_t)mc * C + n) * m2l);
                    }
                }
                next_x();
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int n = n_w0 + q * 32;
                    const f16x8 bvv = *(const f16x8*)(p.b2 + n);
                    const f16x8 rbv = *(const f16x8*)(rbp + (size_t)(g * p.ld_rowbias + n) * mrb);
                    const f32x4 a0 = acc2[rf][2 * q], a1 = acc2[rf][2 * q + 1];
                    const float vv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    f16x8 o;
                    int s8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float xv = (vv[e] + (float)bvv[e] + (float)rbv[e]) * p.c_acc;
                        if (LO) xv += p.c_r1 * ew_split_dec(q1v[q][e], ew_sbyte(q1l[q][e >> 2], e & 3));
                        else xv += p.c_r1 * (float)q1v[q][e];
                        if (R2) {
                            if (LO) xv += p.c_r2 * ew_split_dec(q2v[q][e], ew_sbyte(q2l[q][e >> 2], e & 3));
                            else xv += p.c_r2 * (float)q2v[q][e];
                        }
                        o[e] = (f16)xv;
                        s8[e] = ew_split_enc(xv, o[e]);
                    }
                    if (m < p.M) {
                        *(f16x8*)(p.out + (size_t)m * C + n) = o;
                        if (LO && p.out_lo)
                            *(u32x2*)(p.out_lo + (size_t)m * C + n) = (u32x2){ew_pack4(s8[0], s8[1], s8[2], s8[3]), ew_pack4(s8[4], s8[5], s8[6], s8[7])};
                    }
                }
            }
        }
        FF_WAIT_VM0();      // next tile's x fragments (and this tile's stores) are complete before the chunk loop starts
    }
    FF_WAIT_VM0();          // the last hand-overs requested images nobody reads: let them land
}

}  // namespace


// Host-side layout contract of the packs (evoworld_amd/ops.py: ff_pack builds them with torch index ops):
//   w1p  [40 chunks][5 K-tiles][64 staged rows][8 slots][8 halves]: staged row r = 16 j + i, j = 2h + vg (vg: 0 value, 1 gate),
//        i = 4 fks + e  <-  proj row (vg ? HID : 0) + 32 c + 8 fks + 4 h + e;   slot sl holds k = 64 kt + 8 (sl ^ (r & 7)) .. +8
//   b1p  [40][64] in the same staged-row order
//   w2p  [40 chunks][320 staged rows][4 slots][8 halves]: staged row r = 16 jj + i  <-  output channel (jj >> 1) * 32 + (i >> 2) * 8 +
//        (jj & 1) * 4 + (i & 3);   slot sl holds hidden k = 32 c + 8 (sl ^ g((r >> 2) & 3)), g = {0, 2, 3, 1} .. +8
extern "C" ew_status ew_ff_geglu320_f16(const ew_ff_args* a, void* stream) {
    EW_REQUIRE(a != nullptr, "ew_ff_geglu320_f16: null args");
    EW_REQUIRE(a->x && a->w1p && a->b1p && a->w2p && a->b2 && a->out && a->zero_page, "ew_ff_geglu320_f16: null pointer");
    EW_REQUIRE(a->C == C && a->hidden == HID, "ew_ff_geglu320_f16: built for C = 320, hidden = 1280 (got %d, %d)", a->C, a->hidden);
    EW_REQUIRE(a->M > 0 && a->rows_per_group >= 1, "ew_ff_geglu320_f16: bad M / rows_per_group");
    EW_REQUIRE(!a->rowbias || a->ld_rowbias % 8 == 0, "ew_ff_geglu320_f16: ld_rowbias must be a multiple of 8");
    EW_REQUIRE((!a->r1_lo || a->r1) && (!a->r2_lo || a->r2), "ew_ff_geglu320_f16: r1_lo / r2_lo need r1 / r2");
    EW_REQUIRE((long long)a->M * C * 2 < (1LL << 40), "ew_ff_geglu320_f16: M too large");
    FfP p;
    p.x = (const f16*)a->x; p.w1p = (const f16*)a->w1p; p.b1p = (const f16*)a->b1p; p.w2p = (const f16*)a->w2p; p.b2 = (const f16*)a->b2;
    p.rowbias = (const f16*)a->rowbias; p.r1 = (const f16*)a->r1; p.r2 = (const f16*)a->r2;
    p.r1_lo = (const int8_t*)a->r1_lo; p.r2_lo = (const int8_t*)a->r2_lo; p.out = (f16*)a->out; p.out_lo = (int8_t*)a->out_lo;
    p.zero_page = (const f16*)a->zero_page;
    p.M = a->M; p.rows_per_group = a->rows_per_group; p.ld_rowbias = a->rowbias ? a->ld_rowbias : 0;
    p.n_tiles = ew_cdiv(a->M, BM);
    p.c_acc = a->c_acc; p.c_r1 = a->r1 ? a->c_r1 : 0.f; p.c_r2 = a->r2 ? a->c_r2 : 0.f;
    const int ncu = ew_cu_budget();                 // 256 unless the caller runs on a CU-masked stream
    const int grid = p.n_tiles < ncu ? p.n_tiles : ncu;
    const bool lo = a->r1_lo || a->r2_lo || a->out_lo;
    const bool r2 = a->r2 != nullptr;
    hipStream_t s = (hipStream_t)stream;
#define FF_LAUNCH(LO_, R2_)                                                                                          \
    do {                                                                                                             \
        static std::atomic<unsigned long long> mask{0};                                                              \
        if (ew_status st = ew_ensure_dynamic_lds((const void*)ff320_kernel<LO_, R2_>, LDS_BYTES, mask)) return st;   \
        hipLaunchKernelGGL((ff320_kernel<LO_, R2_>), dim3(grid), dim3(64 * NWV), LDS_BYTES, s, p);                   \
    } while (0)
    if (lo && r2) FF_LAUNCH(true, true);
    else if (lo) FF_LAUNCH(true, false);
    else if (r2) FF_LAUNCH(false, true);
    else FF_LAUNCH(false, false);
#undef FF_LAUNCH
    return ew_check_launch("ew_ff_geglu320_f16");
}
