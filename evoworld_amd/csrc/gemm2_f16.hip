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
//   * swapped MFMA operands (D = W_frag x A_frag^T): every lane ends up with 4 consecutive output columns of one row,
//     so the epilogue (bias / row-bias / SiLU / GEGLU / two residuals) runs straight from registers with 8-byte loads
//     and stores -- no LDS patch, which is what lets the ring keep streaming during the epilogue.
// Same argument block, same addressing modes (dense / conv3x3 / temporal 3-tap, dual source, zero page) as gen 1.
#include "gemm_common.h"

namespace {

#define EW_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// gfx9 s_waitcnt simm16: vmcnt[3:0]=bits3:0, expcnt=bits6:4, lgkmcnt=bits11:8, vmcnt[5:4]=bits15:14.  The BUILTIN form is
// used for lgkmcnt so that hipcc's own waitcnt model knows the LDS queue is empty (an inline-asm wait is opaque to it).
#define EW_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define EW_COMPILER_FENCE() asm volatile("" ::: "memory")

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) EW_WAIT_VMCNT(0);
    else if constexpr (N == 1) EW_WAIT_VMCNT(1);
    else if constexpr (N == 2) EW_WAIT_VMCNT(2);
    else if constexpr (N == 3) EW_WAIT_VMCNT(3);
    else if constexpr (N == 4) EW_WAIT_VMCNT(4);
    else if constexpr (N == 5) EW_WAIT_VMCNT(5);
    else if constexpr (N == 6) EW_WAIT_VMCNT(6);
    else if constexpr (N == 7) EW_WAIT_VMCNT(7);
    else if constexpr (N == 8) EW_WAIT_VMCNT(8);
    else static_assert(N <= 8, "extend wait_vmcnt");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE>
__global__ __launch_bounds__(512, 2) void gemm2_kernel(const GemmP p) {
    constexpr int BK = 64;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int A_GROUPS = BM / 8, B_GROUPS = BN / 8;
    constexpr int GA = A_GROUPS / 8;                       // A row-groups per wave (A_GROUPS is a multiple of 8)
    constexpr int GB = (B_GROUPS + 7) / 8;                 // W row-groups per wave (the last may be partial)
    constexpr int GB_FULL = B_GROUPS / 8;                  // W row-groups every wave owns
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int A_BYTES = BM * 128;
    static_assert(WAVES_M * WAVES_N == 8 && WM == 64, "8 waves, 64-row wave tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const bool has_tail = (GB > GB_FULL) && (wave + 8 * GB_FULL < B_GROUPS);   // owns the partial last W group
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

    // ---------------- loader state (runs 2 K-tiles ahead of the MFMA stream, across output tiles) ----------------
    int ld_i = 0, ld_kt = 0, ld_tap = 0, ld_cc = 0, ld_key = -1;
    int a_y[GA], a_x[GA], a_img[GA], a_row[GA];
    const f16* a_src[GA];
    bool a_ok[GA];
    const f16* b_ptr[GB];

    auto loader_new_tile = [&]() {
        const int id = ld_i * G + seq0;
        const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            int m = m0 + (wave + 8 * i) * 8 + srow;
            m = m < p.M ? m : p.M - 1;
            a_row[i] = m;
            if constexpr (MODE == EW_A_CONV3X3) {
                const int hw = p.h_out * p.w_out;
                const int img = m / hw, rem = m - img * hw;
                a_img[i] = img; a_y[i] = rem / p.w_out; a_x[i] = rem - a_y[i] * p.w_out;
            } else if constexpr (MODE == EW_A_CONVT3) {
                const int tp = p.tT * p.tP;
                const int b = m / tp, rem = m - b * tp;
                a_img[i] = b; a_y[i] = rem / p.tP; a_x[i] = rem - a_y[i] * p.tP;
            } else {
                a_img[i] = 0; a_y[i] = 0; a_x[i] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            int n = n0 + (wave + 8 * j) * 8 + srow;
            n = n < p.N ? n : p.N - 1;
            b_ptr[j] = p.w + (size_t)n * p.K + slot * 8;
        }
        ld_kt = 0; ld_tap = 0; ld_cc = 0; ld_key = -1;
    };

    auto stage = [&](char* buf) {   // issue the DMA of the next K-tile of the stream into ring slot `buf`
        if (ld_kt == 0) loader_new_tile();
        const int tap = ld_tap, cc = ld_cc;
        const bool second = cc >= p.c1;
        const int key = tap * 2 + (second ? 1 : 0);
        if (key != ld_key) {
            ld_key = key;
            const f16* base = second ? p.a2 : p.a;
            const int ld = second ? p.lda2 : p.lda;
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                long long pix;
                if constexpr (MODE == EW_A_CONV3X3) {
                    const int ky = tap / 3, kx = tap - ky * 3;
                    int iy = a_y[i] * p.stride + ky - 1, ix = a_x[i] * p.stride + kx - 1;
                    const int hlim = p.upsample ? 2 * p.h_in : p.h_in, wlim = p.upsample ? 2 * p.w_in : p.w_in;
                    const bool ok = iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
                    if (p.upsample) { iy >>= 1; ix >>= 1; }
                    pix = ok ? ((long long)a_img[i] * p.h_in + iy) * p.w_in + ix : -1;
                } else if constexpr (MODE == EW_A_CONVT3) {
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
        for (int i = 0; i < GA; ++i) glds16(a_src[i] + (a_ok[i] ? ch : 0), buf + (wave + 8 * i) * 1024);
        const size_t koff = (size_t)ld_kt * BK;
#pragma unroll
        for (int j = 0; j < GB_FULL; ++j) glds16(b_ptr[j] + koff, buf + A_BYTES + (wave + 8 * j) * 1024);
        if constexpr (GB > GB_FULL) {
            if (has_tail) glds16(b_ptr[GB - 1] + koff, buf + A_BYTES + (wave + 8 * (GB - 1)) * 1024);
        }
        ld_cc += BK;
        if (ld_cc == C) { ld_cc = 0; ++ld_tap; }
        if (++ld_kt == nk) { ld_kt = 0; ++ld_i; }
    };

    auto wait_landed = [&](bool more_in_flight) {   // own DMA of the NEXT K-tile landed; leave the newest tile in flight
        if (!more_in_flight) { wait_vmcnt<0>(); return; }
        if constexpr (GB > GB_FULL) {
            if (has_tail) wait_vmcnt<GA + GB_FULL + 1>(); else wait_vmcnt<GA + GB_FULL>();
        } else {
            wait_vmcnt<GA + GB_FULL>();
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
    auto read_frags = [&](const char* buf, int so, f16x8 (&af)[FM], f16x8 (&bf)[FN]) {
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

    // ---------------- prologue: two K-tiles in flight, first fragments in registers ----------------
    stage(smem);
    if (V > 1) stage(smem + STAGE);
    wait_landed(V > 1);
    EW_COMPILER_FENCE();
    __builtin_amdgcn_s_barrier();
    EW_COMPILER_FENCE();
    read_frags(smem, so0, af0, bf0);

    int cur_i = 0, cur_kt = 0;
    int s_cur = 0;                                   // ring slot of stream position v
    for (int v = 0; v < V; ++v) {
        const int s_nxt = s_cur == 2 ? 0 : s_cur + 1;
        const int s_nn = s_nxt == 2 ? 0 : s_nxt + 1;
        const char* cur = smem + s_cur * STAGE;
        if (v + 2 < V) stage(smem + s_nn * STAGE);   // slot of v-1: every wave passed barrier(v-1) after its last read
        // ---- half-step 0: MFMA on k[0,32), fetch fragments of k[32,64)
        EW_WAIT_LGKM0();   // af0/bf0 (read one half-step ago) have landed: free, and it lets the MFMAs below start
                           // without waiting for the reads issued next (hipcc otherwise emits lgkmcnt(0) after them)
        read_frags(cur, so1, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads AHEAD of the MFMAs (hipcc otherwise sinks them to the end)
        mma(af0, bf0);
        // ---- publish K-tile v+1.  Unconditional (also on the last position, where the fragments read from the ring are
        // stale and never used): a conditional here makes hipcc put a conservative lgkmcnt(0) at the join, in front of
        // the half-step-1 MFMAs, which would expose the LDS latency of the reads just issued.
        wait_landed(v + 2 < V);
        EW_WAIT_LGKM0();
        EW_COMPILER_FENCE();
        __builtin_amdgcn_s_barrier();
        EW_COMPILER_FENCE();
        read_frags(smem + s_nxt * STAGE, so0, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- half-step 1
        mma(af1, bf1);
        s_cur = s_nxt;
        if (++cur_kt == nk) {
            // ------------------------- epilogue of output tile cur_i (registers only) -------------------------
            cur_kt = 0;
            const int id = cur_i * G + seq0;
            ++cur_i;
            const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
            const int mb = tm * BM + wm * WM + frow;
            const int nb = tn * BN + wn * WN + fks * 4;
            if (p.act != EW_ACT_GEGLU) {
                // Branch-free operand fetch: an absent bias / row-bias / residual reads a zero page with stride 0, so the
                // compiler sees straight-line code and emits COUNTED vmcnt waits (a uniform branch per operand made it
                // fall back to vmcnt(0) after every store).  Loads of column fragment j+1 are issued before the stores
                // of fragment j (software pipeline), so a wait never has to drain the stores.
                const f16* rbp = p.rowbias ? p.rowbias : p.zero_page;
                const f16* r1p = p.r1 ? p.r1 : p.zero_page;
                const f16* r2p = p.r2 ? p.r2 : p.zero_page;
                const f16* bp = p.bias ? p.bias : p.zero_page;
                const int ldrb = p.rowbias ? p.ld_rowbias : 0, ld1 = p.r1 ? p.ld_r1 : 0, ld2 = p.r2 ? p.ld_r2 : 0;
                const int mrb = p.rowbias ? 1 : 0, m1 = p.r1 ? 1 : 0, m2 = p.r2 ? 1 : 0, mbias = p.bias ? 1 : 0;
                const int N4 = p.N - 4;
                size_t orow[FM];
                size_t rrow1[FM], rrow2[FM], rrowb[FM];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int mc = min(mb + i * 16, p.M - 1);
                    orow[i] = (size_t)mc * p.ld_out;
                    rrow1[i] = (size_t)mc * ld1;
                    rrow2[i] = (size_t)mc * ld2;
                    rrowb[i] = (size_t)(mc / p.rows_per_group) * ldrb;
                }
                f16x4 rb[2][FM], q1[2][FM], q2[2][FM], bb[2];
                auto fetch = [&](int j, int set) {
                    const int n = nb + j * 16;
                    const int nc = n < p.N ? n : N4;
                    bb[set] = *(const f16x4*)(bp + nc * mbias);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        rb[set][i] = *(const f16x4*)(rbp + rrowb[i] + nc * mrb);
                        q1[set][i] = *(const f16x4*)(r1p + rrow1[i] + nc * m1);
                        q2[set][i] = *(const f16x4*)(r2p + rrow2[i] + nc * m2);
                    }
                };
                fetch(0, 0);
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int set = j & 1;
                    if (j + 1 < FN) fetch(j + 1, set ^ 1);
                    const int n = nb + j * 16;
                    const f32x4 bv = {(float)bb[set][0], (float)bb[set][1], (float)bb[set][2], (float)bb[set][3]};
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int m = mb + i * 16;
                        f32x4 val = acc[i][j] + bv;
                        val += (f32x4){(float)rb[set][i][0], (float)rb[set][i][1], (float)rb[set][i][2], (float)rb[set][i][3]};
                        if (p.act == EW_ACT_SILU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = ew_silu(val[e]);
                        }
                        val *= p.c_acc;
                        val += p.c_r1 * (f32x4){(float)q1[set][i][0], (float)q1[set][i][1], (float)q1[set][i][2], (float)q1[set][i][3]};
                        val += p.c_r2 * (f32x4){(float)q2[set][i][0], (float)q2[set][i][1], (float)q2[set][i][2], (float)q2[set][i][3]};
                        if (m < p.M && n < p.N)
                            *(f16x4*)(p.out + orow[i] + n) = (f16x4){(f16)val[0], (f16)val[1], (f16)val[2], (f16)val[3]};
                        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            } else {
                if constexpr (FN % 2 == 0) {
#pragma unroll
                    for (int q = 0; q < FN / 2; ++q) {
                        const int ns = nb + q * 32;                      // staged column of the value fragment
                        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
                        if (p.bias && ns < p.N) {
                            const f16x4 b0 = *(const f16x4*)(p.bias + ns);
                            const f16x4 b1 = *(const f16x4*)(p.bias + ns + 16);
                            bv = (f32x4){(float)b0[0], (float)b0[1], (float)b0[2], (float)b0[3]};
                            bg = (f32x4){(float)b1[0], (float)b1[1], (float)b1[2], (float)b1[3]};
                        }
                        const int no = ((tn * BN + wn * WN) >> 1) + q * 16 + fks * 4;
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            const int m = mb + i * 16;
                            if (m < p.M && ns < p.N) {
                                const f32x4 vv = acc[i][2 * q] + bv, gg = acc[i][2 * q + 1] + bg;
                                f16x4 o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = (f16)(vv[e] * ew_gelu(gg[e]));
                                *(f16x4*)(p.out + (size_t)m * p.ld_out + no) = o;
                            }
                            acc[i][2 * q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            acc[i][2 * q + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE>
ew_status launch2(const GemmP& p, hipStream_t s) {
    GemmP q = p;
    q.tiles_m = ew_cdiv(p.M, BM);
    q.tiles_n = ew_cdiv(p.N, BN);
    const size_t lds = 3 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm2_kernel<BM, BN, WAVES_M, WAVES_N, MODE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ew_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return EW_ERR_HIP; }
        attr_set = true;
    }
    const long long tiles = (long long)q.tiles_m * q.tiles_n;
    if (tiles <= 0 || tiles > 0x7fffffffLL) { ew_set_error("ew_gemm_f16: bad grid"); return EW_ERR_INVALID_ARG; }
    int grid = 256;                                   // one persistent workgroup per CU (MI355X: 256 CUs)
    if (tiles < grid) grid = (int)((tiles + 7) / 8 * 8);
    hipLaunchKernelGGL((gemm2_kernel<BM, BN, WAVES_M, WAVES_N, MODE>), dim3(grid), dim3(512), lds, s, q);
    return ew_check_launch("ew_gemm_f16(gen2)");
}

template <int MODE>
ew_status dispatch_mode(const GemmP& p, hipStream_t s) {
    if (p.act != EW_ACT_GEGLU && p.N % 160 == 0) return launch2<256, 160, 4, 2, MODE>(p, s);
    return launch2<128, 256, 2, 4, MODE>(p, s);
}

}  // namespace

ew_status ew_gemm2_dispatch(const GemmP& p, hipStream_t s) {
    if (p.mode == EW_A_CONV3X3) return dispatch_mode<EW_A_CONV3X3>(p, s);
    if (p.mode == EW_A_CONVT3) return dispatch_mode<EW_A_CONVT3>(p, s);
    return dispatch_mode<EW_A_DENSE>(p, s);
}
