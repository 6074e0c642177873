// conv_small_n.hip -- stride-1 "same" 3x3 convolution with a handful of output channels (N <= 16), gfx950 (round 6).
//
// The U-Net's conv_out (320 -> 4 channels on 2*T*72*128 = 460800 pixels; evoworld/trainer/unet_plucker.py:239-244, 480) -- and, with the split
// operands of round 5 / 6, three K blocks [x_hi | x_lo] x [W_hi | W_hi] + x_hi x W_lo = 8640 deep -- is 64 GFLOP of useful work against 5.3 GB of
// activation bytes through the nine taps.  On the generic tile kernels it ran on a 256x160 tile with 156 of 160 columns padding: 1.39 ms, 8 x its
// byte floor.  Here the tile is what the problem is: no N tiling at all.
//   * v_mfma_f32_16x16x32_f16 with the operand roles of the big kernels: the weight fragment (16 output channels x 32 k, rows >= N are zeros) is
//     the A operand, a fragment of 16 consecutive pixels x 32 input channels the B operand -> D[n][pixel]: lanes 0..15 own the N <= 4 real channels
//     of one pixel each (one 8-byte store per pixel), lanes 16..63 own padding;
//   * a wave owns 32 consecutive output pixels (2 fragments, 8 accumulator registers; 70 VGPRs: four waves per SIMD -- with 4 fragments the
//     kernel needs 134, i.e. three) and walks K in the packed weight order
//     [64-channel chunk][tap][64] (ops.pack_conv_weight): per 32-deep k-step ONE ds_read_b128 of the weight fragment serves two MFMAs; the
//     activation fragments are 16-byte global loads straight into the MFMA operand registers (lane = (pixel, 8-channel slice)): no LDS on
//     that side -- the nine taps of a chunk re-hit the same lines in the CU's vector cache / L2 (the 64-byte halves of a 128-byte line are
//     consecutive k-steps), padding taps read the zero page;
//   * the whole weight matrix (N rows x K, 69 KB at N = 4, K = 8640) sits in LDS for the lifetime of the workgroup, rows padded by 16 bytes so
//     that the four real rows of a fragment read fall into distinct banks; lanes of the zero rows read one shared zero slot;
//   * 512 threads = 8 waves per workgroup = 256 pixels, two workgroups per CU (LDS) = 16 waves per CU to hide the load latency.
// Measured (tools/bench_conv_out.py, conv_out's forward shape): 1.39 ms on generation 2's 256x160 tile -> 1.00 ms -> 0.66 ms with the twin K block
// (below).  1, 2 or 4 fragments per wave and 2-4 waves per SIMD all land within 4 % of each other: time is proportional to the activation bytes
// requested (3.5 GB of 16-byte row pieces at ~5.4 TB/s), not to MFMA or LDS work.
// Dispatched by ew_gemm_f16 for mode conv3x3, N <= 16, stride 1, no upsample / shift, bias only (conv_small_n_wants).
#include "gemm_common.h"

namespace {

constexpr int WAVES = 8, FR = 2;                 // fragments of 16 pixels per wave
constexpr int PIX_PER_WG = WAVES * FR * 16;      // 256

template <int NR>                                // NR = real output channels rounded up to 4 (4 | 8 | 12 | 16): rows kept in LDS
__global__ __launch_bounds__(64 * WAVES, 4) void conv_small_n_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;
    const int row_bytes = K * 2 + 16;            // + 16: rows n and n + 2 would otherwise share banks (K * 2 = 17280 = 0 mod 256 * ... at K = 8640)
    char* const zero_slot = smem + NR * row_bytes;
    // ---- weights -> LDS (once per workgroup): rows 0 .. N-1 of W [N, K], the rest zeros
    for (int i = tid; i < NR * (K / 8); i += 64 * WAVES) {
        const int n = i / (K / 8), kk = i - n * (K / 8);
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < p.N) v = *(const f16x8*)(p.w + (size_t)n * K + kk * 8);
        *(f16x8*)(smem + n * row_bytes + kk * 16) = v;
    }
    if (tid < 4) *(f32x4*)(zero_slot + tid * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int frow = lane & 15, fks = lane >> 4;                 // fragment row (pixel / output channel) and 8-wide k slice of this lane
    // weight fragment address: row frow (zero slot for rows >= NR), advancing 64 bytes per k-step; the zero lanes do not advance
    const char* wb = frow < NR ? smem + frow * row_bytes + fks * 16 : zero_slot + fks * 16;
    const int wstep = frow < NR ? 64 : 0;

    const int HW = p.h_out * p.w_out;
    const long long m_base = ((long long)blockIdx.x * WAVES + wave) * (FR * 16);
    // per fragment: this lane's pixel, its tap validity mask and its centre pixel index (stride 1, same geometry in and out)
    int ctr[FR], mask[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        long long m = m_base + f * 16 + frow;
        if (m >= p.M) m = p.M - 1;
        const int img = (int)(m / HW), rem = (int)(m - (long long)img * HW);
        const int y = rem / p.w_out, x = rem - y * p.w_out;
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            if (iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in) mk |= 1 << t;
        }
        mask[f] = mk;
        ctr[f] = (int)m;                                          // stride 1, no upsample: input pixel index == output pixel index
    }
    f32x4 acc[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16* const zp = p.zero_page + fks * 8;

    int wk = 0;                                                   // byte offset of the current k-step inside a weight row
    // Split-operand form (conv_out): source 2 is the first c2 channels of source 1 AGAIN (same tensor, same row stride) against a second weight
    // block (W_lo).  Those chunks are not re-loaded: chunk ch < c2 / 64 of source 1 feeds two MFMAs per k-step, with the weight fragments of K
    // blocks ch and c1 / 64 + ch -- a third of the activation loads gone.  (The sum order over K changes; fp32 accumulation.)
    const bool twin = p.c2 > 0 && p.a2 == p.a && p.lda2 == p.lda && p.c2 <= p.c1;
    const int n_chunks = (twin ? p.c1 : p.c1 + p.c2) / 64;
    const int twin_chunks = twin ? p.c2 / 64 : 0;
    const int wk2 = (p.c1 / 64) * 9 * 2 * wstep;                  // byte distance from K block ch to its twin c1 / 64 + ch
    for (int ch = 0; ch < n_chunks; ++ch) {
        const int cc = ch * 64;
        const bool second = cc >= p.c1;
        const f16* base = second ? p.a2 : p.a;
        const int ld = second ? p.lda2 : p.lda;
        const int c0 = (second ? cc - p.c1 : cc) + fks * 8;
        const bool tw = ch < twin_chunks;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dpix = (t / 3 - 1) * p.w_in + (t % 3 - 1);
            f16x8 bf[2][FR];
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const f16* src = ((mask[f] >> t) & 1) ? base + (long long)(ctr[f] + dpix) * ld + c0 : zp;
                const int h = ((mask[f] >> t) & 1) ? 32 : 0;      // second k-half of the chunk: + 32 channels (the zero page does not advance)
                bf[0][f] = *(const f16x8*)src;
                bf[1][f] = *(const f16x8*)(src + h);
            }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const f16x8 wf = *(const f16x8*)(wb + wk);
#pragma unroll
                for (int f = 0; f < FR; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, bf[kh][f], acc[f], 0, 0, 0);   // D[n][pixel]
                if (tw) {
                    const f16x8 wf2 = *(const f16x8*)(wb + wk + wk2);
#pragma unroll
                    for (int f = 0; f < FR; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf2, bf[kh][f], acc[f], 0, 0, 0);
                }
                wk += wstep;
            }
        }
    }
    // ---- epilogue: lane (fks, frow) holds channels 4 fks .. 4 fks + 3 of pixel frow; bias, round, one 8-byte store per pixel and channel quad
    if (fks * 4 < p.N) {
        f16x4 b4 = {0, 0, 0, 0};
        if (p.bias) b4 = *(const f16x4*)(p.bias + fks * 4);
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const long long m = m_base + f * 16 + frow;
            if (m < p.M) {
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)(p.c_acc * (acc[f][e] + (float)b4[e]));
                *(f16x4*)(p.out + m * p.ld_out + fks * 4) = o;
            }
        }
    }
}

}  // namespace

// true when the problem is a stride-1 "same" 3x3 conv with at most 16 output channels and a bias-only epilogue
bool ew_conv_small_n_wants(const GemmP& p) {
    return p.mode == EW_A_CONV3X3 && p.N <= 16 && p.N % 4 == 0 && p.stride == 1 && !p.upsample && !p.conv_shift && p.h_in == p.h_out && p.w_in == p.w_out &&
           !p.rowbias && !p.r1 && !p.r2 && !p.out_lo && p.act == EW_ACT_NONE && p.M >= 4096 && ((p.N + 3) / 4 * 4) * (p.K * 2 + 16) + 64 <= 80 * 1024;
}

extern char g_gemm_last_kernel[64];

ew_status ew_conv_small_n_dispatch(const GemmP& p, hipStream_t s) {
    const int NR = (p.N + 3) / 4 * 4;
    const int lds = NR * (p.K * 2 + 16) + 64;
    const int grid = ew_cdiv(p.M, PIX_PER_WG);
    snprintf(g_gemm_last_kernel, 64, "conv_small_n_kernel<%d>", NR);
#define CSN_LAUNCH(NR_)                                                                                                 \
    do {                                                                                                                \
        static std::atomic<unsigned long long> mask{0};                                                                 \
        if (ew_status st = ew_ensure_dynamic_lds((const void*)conv_small_n_kernel<NR_>, lds, mask)) return st;          \
        hipLaunchKernelGGL(conv_small_n_kernel<NR_>, dim3(grid), dim3(64 * WAVES), lds, s, p);                           \
    } while (0)
    if (NR == 4) CSN_LAUNCH(4);
    else if (NR == 8) CSN_LAUNCH(8);
    else if (NR == 12) CSN_LAUNCH(12);
    else CSN_LAUNCH(16);
#undef CSN_LAUNCH
    return ew_check_launch("ew_gemm_f16(conv_small_n)");
}
