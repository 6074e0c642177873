// gemm_f16.hip -- fused MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
// One kernel family covers every dense contraction of the U-Net denoise step:
// Linear layers, 3x3 spatial convs (stride 1/2, fused nearest-x2 upsample, fused skip-concat),
// the (3,1,1) temporal convs, 1x1 shortcuts -- with bias / time-embedding row-bias / GEGLU /
// residual / AlphaBlender epilogues.  Reference call sites: evoworld/trainer/unet_plucker.py:126-244
// (the diffusers blocks it instantiates).
//
// Design (MI355X-first):
//   * tile BM x BN x 64, 256 threads = 4 wave64 in a 2x2 grid, v_mfma_f32_16x16x32_f16, fp32 accumulate.
//   * A and W tiles go HBM -> LDS with `global_load_lds_dwordx4` (16 B/lane, no VGPR round trip).  The LDS
//     image is lane-linear (DMA constraint), so the bank-conflict-free XOR swizzle (16-B slot ^= row&7) is
//     applied on the per-lane SOURCE address and again on the ds_read_b128 address.
//   * the conv gather (im2col) is done by the DMA source address itself: each lane points at the input
//     pixel of its output row for the current tap, or at a zero page for padding -- no im2col buffer, no
//     concat buffer (two source tensors), no upsample buffer.
//   * double-buffered LDS, ONE barrier per K-tile: next tile's DMA is issued before the MFMAs of the
//     current tile.
//   * XCD-aware tile order: consecutive tile ids (N fastest) land on the same XCD so the A panel is an L2 hit.
//   * epilogue: accumulators -> wave-private LDS patch -> row-major float4 -> fused ops -> 8-byte fp16 stores.
#include "gemm_common.h"
#include <stdlib.h>

namespace {

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP p) {
    constexpr int BK = 64;                 // fp16 elements per K-tile (128 B rows)
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int GA = BM / 32;            // 8-row DMA groups per wave for A
    constexpr int GB = BN / 32;            // ... for W
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware bijective tile remap (guide T1): blocks b, b+8, b+16.. share an XCD ----
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane staging geometry: lane -> (row within 8-row group, physical 16-B slot) ----
    const int srow = lane >> 3;            // 0..7
    const int slot = (lane & 7) ^ srow;    // logical slot fetched into physical slot lane&7
    const int C = p.c1 + p.c2;
    const int tiles_per_tap = C / BK;
    const int nk = p.K / BK;

    // A rows handled by this lane: groups g = wave*GA + i, row = 8g + srow
    int a_row[GA];                         // clamped global row m
    int a_y[GA], a_x[GA], a_img[GA];       // conv geometry (mode 1: oy, ox, img; mode 2: t, p, b)
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        int m = m0 + (wave * GA + i) * 8 + srow;
        m = m < p.M ? m : p.M - 1;
        a_row[i] = m;
        if (p.mode == EW_A_CONV3X3) {
            const int hw = p.h_out * p.w_out;
            const int img = m / hw, rem = m - img * hw;
            a_img[i] = img; a_y[i] = rem / p.w_out; a_x[i] = rem - a_y[i] * p.w_out;
        } else if (p.mode == EW_A_CONVT3) {
            const int tp = p.tT * p.tP;
            const int b = m / tp, rem = m - b * tp;
            a_img[i] = b; a_y[i] = rem / p.tP; a_x[i] = rem - a_y[i] * p.tP;
        } else {
            a_img[i] = 0; a_y[i] = 0; a_x[i] = 0;
        }
    }
    const f16* b_ptr[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        int n = n0 + (wave * GB + j) * 8 + srow;
        n = n < p.N ? n : p.N - 1;
        b_ptr[j] = p.w + (size_t)n * p.K + slot * 8;
    }

    const f16* a_src[GA];                  // source row pointer (incl. slot) for the current (tap, source)
    bool a_ok[GA];                         // false -> padding (zero page)
    int st_tap = 0, st_cc = 0;             // running (tap, channel) of the next tile to stage
    int cur_key = -1;

    auto stage = [&](int kt, char* buf) {
        const int tap = st_tap, cc = st_cc;
        const int ntap = p.mode == EW_A_CONV3X3 ? 9 : (p.mode == EW_A_CONVT3 ? 3 : 1);   // chunk-major, tap-minor K order
        if (++st_tap == ntap) { st_tap = 0; st_cc += BK; }
        const bool second = cc >= p.c1;
        const int key = tap * 2 + (second ? 1 : 0);
        if (key != cur_key) {              // wave-uniform branch: new tap or switch to the concat source
            cur_key = key;
            const f16* base = second ? p.a2 : p.a;
            const int ld = second ? p.lda2 : p.lda;
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                long long pix;
                if (p.mode == EW_A_CONV3X3) {
                    const int ky = tap / 3, kx = tap - ky * 3;
                    int iy = a_y[i] * p.stride + ky - 1 + p.conv_shift, ix = a_x[i] * p.stride + kx - 1 + p.conv_shift;
                    const int hlim = p.upsample ? 2 * p.h_in : p.h_in, wlim = p.upsample ? 2 * p.w_in : p.w_in;
                    const bool ok = iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
                    if (p.upsample) { iy >>= 1; ix >>= 1; }
                    pix = ok ? ((long long)a_img[i] * p.h_in + iy) * p.w_in + ix : -1;
                } else if (p.mode == EW_A_CONVT3) {
                    const int t = a_y[i] + tap - 1;
                    pix = (t >= 0 && t < p.tT) ? ((long long)a_img[i] * p.tT + t) * p.tP + a_x[i] : -1;
                } else {
                    pix = a_row[i];
                }
                a_ok[i] = pix >= 0;
                a_src[i] = (pix >= 0 ? base + pix * ld : p.zero_page) + slot * 8;
            }
        }
        const int ch = second ? cc - p.c1 : cc;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            glds16(a_src[i] + (a_ok[i] ? ch : 0), buf + (wave * GA + i) * 1024);
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            glds16(b_ptr[j] + (size_t)kt * BK, buf + A_BYTES + (wave * GB + j) * 1024);
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes within a tile): row r, logical slot s -> r*128 + ((s ^ (r&7))<<4)
    const int frow = lane & 15, fks = lane >> 4;
    int a_off[FM], b_off[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a_off[i] = (wm * WM + i * 16 + frow) * 128;
#pragma unroll
    for (int j = 0; j < FN; ++j) b_off[j] = A_BYTES + (wn * WN + j * 16 + frow) * 128;
    const int sw = frow & 7;   // (row & 7): tile-row offsets are multiples of 16

    stage(0, smem);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * BUF;
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * BUF);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int so = ((kk * 4 + fks) ^ sw) << 4;
            f16x8 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(cur + a_off[i] + so);
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(cur + b_off[j] + so);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ------------------------------- epilogue -------------------------------
    constexpr int LDP = WN + 4;                       // padded row (floats) of the wave-private patch
    float* patch = (float*)smem + wave * (16 * LDP);
    const bool geglu = p.act == EW_ACT_GEGLU;
    const int wcol0 = n0 + wn * WN;                   // first (staged) column of this wave
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) patch[(fks * 4 + r) * LDP + j * 16 + frow] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
        const int mrow0 = m0 + wm * WM + i * 16;
        if (!geglu) {
            constexpr int VPR = WN / 4;
#pragma unroll
            for (int it = 0; it < (16 * VPR + 63) / 64; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / VPR, vc = (idx - row * VPR) * 4;
                const int m = mrow0 + row, n = wcol0 + vc;
                if (idx < 16 * VPR && m < p.M && n < p.N) {
                    f32x4 v = *(const f32x4*)(patch + row * LDP + vc);
                    if (p.bias) {
                        const f16x4 b = *(const f16x4*)(p.bias + n);
                        v += (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                    }
                    if (p.rowbias) {
                        const f16x4 b = *(const f16x4*)(p.rowbias + (size_t)(m / p.rows_per_group) * p.ld_rowbias + n);
                        v += (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                    }
                    if (p.act == EW_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ew_silu(v[e]);
                    } else if (p.act == EW_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ew_gelu(v[e]);
                    }
                    v *= p.c_acc;
                    if (p.r1) {
                        const f16x4 b = *(const f16x4*)(p.r1 + (size_t)m * p.ld_r1 + n);
                        f32x4 bf = {(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                        if (p.r1_lo) {
                            const unsigned l = *(const unsigned*)(p.r1_lo + (size_t)m * p.ld_r1 + n);
                            bf = (f32x4){ew_split_dec(b[0], ew_sbyte(l, 0)), ew_split_dec(b[1], ew_sbyte(l, 1)),
                                         ew_split_dec(b[2], ew_sbyte(l, 2)), ew_split_dec(b[3], ew_sbyte(l, 3))};
                        }
                        v += p.c_r1 * bf;
                    }
                    if (p.r2) {
                        const f16x4 b = *(const f16x4*)(p.r2 + (size_t)m * p.ld_r2 + n);
                        f32x4 bf = {(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                        if (p.r2_lo) {
                            const unsigned l = *(const unsigned*)(p.r2_lo + (size_t)m * p.ld_r2 + n);
                            bf = (f32x4){ew_split_dec(b[0], ew_sbyte(l, 0)), ew_split_dec(b[1], ew_sbyte(l, 1)),
                                         ew_split_dec(b[2], ew_sbyte(l, 2)), ew_split_dec(b[3], ew_sbyte(l, 3))};
                        }
                        v += p.c_r2 * bf;
                    }
                    const f16x4 oh = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    *(f16x4*)(p.out + (size_t)m * p.ld_out + n) = oh;
                    if (p.out_lo)
                        *(unsigned*)(p.out_lo + (size_t)m * p.ld_out + n) =
                            ew_pack4(ew_split_enc(v[0], oh[0]), ew_split_enc(v[1], oh[1]), ew_split_enc(v[2], oh[2]), ew_split_enc(v[3], oh[3]));
                }
            }
        } else {
            // staged columns come in blocks of 32 = [16 value | 16 gate]; output has N/2 columns
            constexpr int VPR = WN / 8;               // output float4 vectors per row
#pragma unroll
            for (int it = 0; it < (16 * VPR + 63) / 64; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / VPR, ov = idx - row * VPR;     // ov: output vector index
                const int q = ov >> 2, c = (ov & 3) * 4;             // block q, column c within 16
                const int m = mrow0 + row;
                const int ns = wcol0 + q * 32 + c;                   // staged column of the value
                if (idx < 16 * VPR && m < p.M && ns < p.N) {
                    f32x4 v = *(const f32x4*)(patch + row * LDP + q * 32 + c);
                    f32x4 g = *(const f32x4*)(patch + row * LDP + q * 32 + 16 + c);
                    if (p.bias) {
                        const f16x4 bv = *(const f16x4*)(p.bias + ns);
                        const f16x4 bg = *(const f16x4*)(p.bias + ns + 16);
                        v += (f32x4){(float)bv[0], (float)bv[1], (float)bv[2], (float)bv[3]};
                        g += (f32x4){(float)bg[0], (float)bg[1], (float)bg[2], (float)bg[3]};
                    }
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[e] * ew_gelu(g[e]);
                    const int no = (wcol0 >> 1) + q * 16 + c;
                    *(f16x4*)(p.out + (size_t)m * p.ld_out + no) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int BM, int BN>
ew_status launch(const GemmP& p, hipStream_t s) {
    GemmP q = p;
    q.tiles_m = ew_cdiv(p.M, BM);
    q.tiles_n = ew_cdiv(p.N, BN);
    const size_t lds = 2 * (BM + BN) * 128;
    static std::atomic<unsigned long long> attr_mask{0};                   // per (kernel instantiation, device)
    if (ew_status st = ew_ensure_dynamic_lds((const void*)gemm_kernel<BM, BN>, (int)lds, attr_mask)) return st;
    const long long nblk = (long long)q.tiles_m * q.tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) { ew_set_error("ew_gemm_f16: bad grid"); return EW_ERR_INVALID_ARG; }
    hipLaunchKernelGGL((gemm_kernel<BM, BN>), dim3((unsigned)nblk), dim3(256), lds, s, q);
    return ew_check_launch("ew_gemm_f16");
}

}  // namespace

ew_status ew_gemm2_dispatch(const GemmP& p, hipStream_t s);   // gemm2_f16.hip
ew_status ew_gemm3_dispatch(const GemmP& p, hipStream_t s);   // gemm3_f16.hip
bool ew_gemm3_wants(const GemmP& p, hipStream_t s);
ew_status ew_gemm3_dispatch_b256(const GemmP& p, hipStream_t s);   // gemm3_f16.hip compiled with EW3_BN=256
bool ew_conv_small_n_wants(const GemmP& p);                        // conv_small_n.hip: 3x3 convs with N <= 16 (conv_out)
ew_status ew_conv_small_n_dispatch(const GemmP& p, hipStream_t s);
bool ew_gemm3_wants_b256(const GemmP& p, hipStream_t s);
static int g_gemm_gen = -1;
char g_gemm_last_kernel[64] = "";          // rocprof-style name of the kernel the last ew_gemm_f16 call launched
extern "C" const char* ew_gemm_last_kernel(void) { return g_gemm_last_kernel; }
static int g_gemm_dbg = 0;
extern "C" void ew_set_gemm_debug(int d) { g_gemm_dbg = d; }
extern "C" void ew_set_gemm_generation(int gen) { g_gemm_gen = gen; }
extern "C" int ew_get_gemm_generation(void) {
    if (g_gemm_gen < 0) g_gemm_gen = 3;
    return g_gemm_gen;
}

extern "C" ew_status ew_gemm_f16(const ew_gemm_args* a, void* stream) {
    EW_REQUIRE(a != nullptr, "ew_gemm_f16: null args");
    EW_REQUIRE(a->a && a->w && a->out && a->zero_page, "ew_gemm_f16: null a/w/out/zero_page");
    EW_REQUIRE(a->M > 0 && a->N > 0, "ew_gemm_f16: M,N must be > 0 (M=%d N=%d)", a->M, a->N);
    EW_REQUIRE(a->c1 > 0 && a->c1 % 64 == 0 && a->c2 >= 0 && a->c2 % 64 == 0,
               "ew_gemm_f16: c1,c2 must be multiples of 64 (c1=%d c2=%d)", a->c1, a->c2);
    EW_REQUIRE(a->c2 == 0 || a->a2, "ew_gemm_f16: c2 > 0 needs a2");
    EW_REQUIRE(a->N % 4 == 0 && a->ld_out % 4 == 0, "ew_gemm_f16: N and ld_out must be multiples of 4");
    EW_REQUIRE(a->lda % 8 == 0 && (a->c2 == 0 || a->lda2 % 8 == 0), "ew_gemm_f16: lda must be a multiple of 8");
    EW_REQUIRE(a->rows_per_group >= 1, "ew_gemm_f16: rows_per_group must be >= 1");
    EW_REQUIRE(!a->r1 || a->ld_r1 % 4 == 0, "ew_gemm_f16: ld_r1 must be a multiple of 4");
    EW_REQUIRE(!a->r2 || a->ld_r2 % 4 == 0, "ew_gemm_f16: ld_r2 must be a multiple of 4");
    int taps = 1;
    if (a->mode == EW_A_CONV3X3) {
        taps = 9;
        EW_REQUIRE(a->n_img > 0 && a->h_in > 0 && a->w_in > 0 && a->h_out > 0 && a->w_out > 0 &&
                       (a->stride == 1 || a->stride == 2) && (a->upsample == 0 || a->upsample == 1),
                   "ew_gemm_f16: bad conv3x3 geometry");
        EW_REQUIRE((long long)a->n_img * a->h_out * a->w_out == a->M, "ew_gemm_f16: M != n_img*h_out*w_out");
        EW_REQUIRE(!(a->upsample && a->stride != 1), "ew_gemm_f16: upsample needs stride 1");
        EW_REQUIRE((a->conv_shift == 0 || a->conv_shift == 1) && !(a->conv_shift && a->upsample), "ew_gemm_f16: conv_shift must be 0 or 1 (not with upsample)");
    } else if (a->mode == EW_A_CONVT3) {
        taps = 3;
        EW_REQUIRE(a->tB > 0 && a->tT > 0 && a->tP > 0 && (long long)a->tB * a->tT * a->tP == a->M,
                   "ew_gemm_f16: M != B*T*P");
    } else {
        EW_REQUIRE(a->mode == EW_A_DENSE, "ew_gemm_f16: unknown mode %d", a->mode);
    }
    EW_REQUIRE(a->act == EW_ACT_NONE || a->act == EW_ACT_SILU || a->act == EW_ACT_GEGLU || a->act == EW_ACT_GELU, "ew_gemm_f16: unknown act %d", a->act);
    EW_REQUIRE(!a->rowbias || a->ld_rowbias % 4 == 0, "ew_gemm_f16: ld_rowbias must be a multiple of 4");
    if (a->act == EW_ACT_GEGLU)
        EW_REQUIRE(a->N % 128 == 0 && !a->rowbias && !a->r1 && !a->r2 && !a->out_lo, "ew_gemm_f16: GEGLU needs N %% 128 == 0 and no residuals");
    EW_REQUIRE((!a->r1_lo || a->r1) && (!a->r2_lo || a->r2), "ew_gemm_f16: r1_lo / r2_lo need r1 / r2");
    GemmP p;
    p.a = (const f16*)a->a; p.a2 = (const f16*)a->a2; p.w = (const f16*)a->w; p.bias = (const f16*)a->bias;
    p.rowbias = (const f16*)a->rowbias; p.r1 = (const f16*)a->r1; p.r2 = (const f16*)a->r2; p.out = (f16*)a->out;
    p.zero_page = (const f16*)a->zero_page;
    p.r1_lo = (const int8_t*)a->r1_lo; p.r2_lo = (const int8_t*)a->r2_lo; p.out_lo = (int8_t*)a->out_lo;
    p.conv_shift = a->conv_shift;
    p.M = a->M; p.N = a->N; p.K = taps * (a->c1 + a->c2);
    p.c1 = a->c1; p.c2 = a->c2; p.lda = a->lda; p.lda2 = a->lda2; p.ld_out = a->ld_out; p.ld_r1 = a->ld_r1; p.ld_r2 = a->ld_r2; p.ld_rowbias = a->ld_rowbias;
    p.mode = a->mode; p.n_img = a->n_img; p.h_in = a->h_in; p.w_in = a->w_in; p.h_out = a->h_out; p.w_out = a->w_out;
    p.stride = a->stride; p.upsample = a->upsample; p.tB = a->tB; p.tT = a->tT; p.tP = a->tP;
    p.rows_per_group = a->rows_per_group; p.act = a->act; p.c_acc = a->c_acc; p.c_r1 = a->c_r1; p.c_r2 = a->c_r2;
    p.tiles_m = p.tiles_n = 0;
    p.band = 0;
    p.dbg = g_gemm_dbg;
    hipStream_t s = (hipStream_t)stream;
    // 3x3 convs with a handful of output channels (conv_out: N = 4) have their own kernel (round 6); generation 1 stays the independent cross-check
    if (ew_get_gemm_generation() >= 2 && ew_conv_small_n_wants(p)) return ew_conv_small_n_dispatch(p, s);
    // generation 3 (256x320 tile) where it applies and fills the chip, generation 2 otherwise
    if (ew_get_gemm_generation() >= 3 && ew_gemm3_wants(p, s)) return ew_gemm3_dispatch(p, s);
    if (ew_get_gemm_generation() >= 3 && ew_gemm3_wants_b256(p, s)) return ew_gemm3_dispatch_b256(p, s);
    if (ew_get_gemm_generation() >= 2) return ew_gemm2_dispatch(p, s);
    // generation 1 tile choice: every channel count of the U-Net is a multiple of 160 (320*k); GEGLU and odd sizes use 128
    if (a->act != EW_ACT_GEGLU && a->N % 160 == 0) { snprintf(g_gemm_last_kernel, 64, "gemm_kernel<128, 160>"); return launch<128, 160>(p, s); }
    snprintf(g_gemm_last_kernel, 64, "gemm_kernel<128, 128>");
    return launch<128, 128>(p, s);
}
