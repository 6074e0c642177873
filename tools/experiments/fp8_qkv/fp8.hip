// fp8.hip -- optional fp8 (OCP e4m3) path for the attention q / k / v projections (BASELINE.json configs[4]: "fp16 U-Net + fp8
// MFMA QKV").  Off by default (EW_QKV_FP8=1 / UNet(qkv_fp8=True)); it changes the numerics contract and carries its own
// stated tolerance (tests/test_gpu_fp8.py).
//   ew_quant_rows_fp8   x fp16 [M,K] -> q fp8 [M,K] + scale fp32 [M], scale = amax(row) / 448  (per-token dynamic scaling)
//   ew_gemm_fp8         out[m][n] = (sum_k qa[m][k] * qw[n][k]) * a_scale[m] * w_scale[n], fp32 accumulation on
//                       v_mfma_f32_16x16x32_fp8_fp8, fp16 out.  128x128 tile, 4 wave64 (2x2), 64-byte K-tiles through a
//                       double-buffered padded LDS tile; swapped operands (D[n][m]) so a lane owns 4 consecutive output columns.
// The non-scaled fp8 MFMA runs at the fp16 rate on gfx950: what this path halves is the operand BYTES (HBM + LDS staging) of
// the projections, which are memory-bound at level 0 (K = 320).
#include "common.h"

namespace {

constexpr float FP8_MAX = 448.0f;

// one wave per row; the row lives in registers (K <= 2048, K % 8 == 0)
template <int NV>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const f16* __restrict__ x, uint8_t* __restrict__ q,
                                                             float* __restrict__ scale, int rows, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int VPP = K / 8;
    f16x8 v[NV];
    float amax = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int vi = lane + k * 64;
        if (vi < VPP) {
            v[k] = *(const f16x8*)(x + (size_t)row * K + vi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[k][e]));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int vi = lane + k * 64;
        if (vi < VPP) {
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[k][0] * inv, (float)v[k][1] * inv, w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[k][2] * inv, (float)v[k][3] * inv, w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[k][4] * inv, (float)v[k][5] * inv, w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[k][6] * inv, (float)v[k][7] * inv, w1, true);
            *(int2*)(q + (size_t)row * K + vi * 8) = make_int2(w0, w1);
        }
    }
}

constexpr int TM = 128, TN = 128, TK = 64, LDSROW = 80;       // 64 data bytes + 16 pad per LDS row
constexpr int STAGE_BYTES = (TM + TN) * LDSROW;               // 20,480 B

__global__ __launch_bounds__(256) void gemm_fp8_kernel(const uint8_t* __restrict__ A, const float* __restrict__ a_scale,
                                                       const uint8_t* __restrict__ W, const float* __restrict__ w_scale,
                                                       f16* __restrict__ out, int M, int N, int K, long long ld_out) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + TN - 1) / TN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const int r = lane & 15, ks = lane >> 4;
    // staging: 256 threads x 2 x 16 B per operand: thread -> (row = tid >> 2 (+64), 16-byte column tid & 3)
    const int srow = tid >> 2, scol = (tid & 3) * 16;
    const uint8_t* ap[2];
    const uint8_t* wp[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ma = min(m0 + srow + 64 * h, M - 1), nb = min(n0 + srow + 64 * h, N - 1);
        ap[h] = A + (size_t)ma * K + scol;
        wp[h] = W + (size_t)nb * K + scol;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = K / TK;
    int4 ra[2], rw[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { ra[h] = *(const int4*)(ap[h] + kt * TK); rw[h] = *(const int4*)(wp[h] + kt * TK); }
    };
    auto lstore = [&](int buf) {
        char* b = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(int4*)(b + (srow + 64 * h) * LDSROW + scol) = ra[h];
            *(int4*)(b + (TM + srow + 64 * h) * LDSROW + scol) = rw[h];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* b = smem + (kt & 1) * STAGE_BYTES;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                      // two 32-wide MFMA k-slices per 64-byte K-tile
            long af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const long*)(b + (wm * 64 + i * 16 + r) * LDSROW + kk * 32 + ks * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const long*)(b + (TM + wn * 64 + j * 16 + r) * LDSROW + kk * 32 + ks * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[j], af[i], acc[i][j], 0, 0, 0);       // D[n][m]
        }
        if (kt + 1 < nk) lstore((kt + 1) & 1);                 // the other buffer was last read in iteration kt-1
        __syncthreads();
    }
    // epilogue: lane holds D[n = 4*ks + e][m = r] of every 16x16 fragment
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r;
        if (m >= M) continue;
        const float sa = a_scale[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + ks * 4;
            if (n + 3 < N) {
                const f32x4 sw = *(const f32x4*)(w_scale + n);
                const f16x4 o = {(f16)(acc[i][j][0] * sa * sw[0]), (f16)(acc[i][j][1] * sa * sw[1]),
                                 (f16)(acc[i][j][2] * sa * sw[2]), (f16)(acc[i][j][3] * sa * sw[3])};
                *(f16x4*)(out + (size_t)m * ld_out + n) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) out[(size_t)m * ld_out + n + e] = (f16)(acc[i][j][e] * sa * w_scale[n + e]);
            }
        }
    }
}

}  // namespace

extern "C" ew_status ew_quant_rows_fp8(const void* x, void* q, float* scale, int rows, int K, void* stream) {
    EW_REQUIRE(x && q && scale && rows > 0 && K > 0 && K % 8 == 0 && K <= 2048, "ew_quant_rows_fp8: need K %% 8 == 0, K <= 2048 (K=%d)", K);
    const int nv = (K / 8 + 63) / 64;
    dim3 grid(ew_cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define QL(NV) hipLaunchKernelGGL(quant_rows_fp8_kernel<NV>, grid, block, 0, s, (const f16*)x, (uint8_t*)q, scale, rows, K)
    if (nv == 1) QL(1); else if (nv == 2) QL(2); else if (nv == 3) QL(3); else QL(4);
#undef QL
    return ew_check_launch("ew_quant_rows_fp8");
}

extern "C" ew_status ew_gemm_fp8(const void* a, const float* a_scale, const void* w, const float* w_scale, void* out, int M,
                                 int N, int K, long long ld_out, void* stream) {
    EW_REQUIRE(a && a_scale && w && w_scale && out, "ew_gemm_fp8: null pointer");
    EW_REQUIRE(M > 0 && N > 0 && K > 0 && K % TK == 0 && N % 4 == 0 && ld_out % 4 == 0 && ld_out >= N, "ew_gemm_fp8: need K %% 64 == 0, N %% 4 == 0 (M=%d N=%d K=%d)", M, N, K);
    EW_REQUIRE((((uintptr_t)a | (uintptr_t)w) & 15) == 0 && ((uintptr_t)w_scale & 15) == 0, "ew_gemm_fp8: operands must be 16-byte aligned");
    const long long tiles = (long long)ew_cdiv(M, TM) * ew_cdiv(N, TN);
    EW_REQUIRE(tiles < 0x7fffffffLL, "ew_gemm_fp8: grid too large");
    hipLaunchKernelGGL(gemm_fp8_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)a, a_scale,
                       (const uint8_t*)w, w_scale, (f16*)out, M, N, K, ld_out);
    return ew_check_launch("ew_gemm_fp8");
}
