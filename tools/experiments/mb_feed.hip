// mb_feed.hip -- stand-alone microbenchmark (round 3): per-CU operand feed rate for L2-resident streams.
// Question: is the ~54 GB/s per CU ceiling of global_load_lds_dwordx4 a property of the LDS-DMA return path, or of
// the address path every vector load shares?  If VGPR-destination loads (global_load_dwordx4) of a fragment-major, fully
// contiguous operand run clearly faster, a GEMM whose W operand bypasses LDS can be fed faster than the current kernels.
//
// One persistent 8-wave workgroup per CU (like gemm3).  Every "K-tile" a workgroup moves 72 KB = 72 wave-instructions of 1 KB:
//   mode 0: all 72 pieces through LDS-DMA (9 per wave), rows of 128 B at a 2560 B stride (the gemm3 loader's pattern)
//   mode 1: all 72 pieces as global_load_dwordx4 -> VGPR, 1 KB contiguous per wave instruction (fragment-major pack)
//   mode 2: all 72 pieces as global_load_dwordx4 -> VGPR, 8 rows x 128 B per wave instruction (row-major operand)
//   mode 3: 32 pieces LDS-DMA (A tile, row pattern) + 40 pieces VGPR contiguous (W tile), i.e. the proposed split
//   mode 4: mode 3 with the VGPR part loaded by every M-wave redundantly (4x: 4(M)x2(N) wave layout without LDS sharing)
//   mode 5: mode 0 with two K-tiles in flight (vmcnt(9) instead of vmcnt(0))
//   mode 6: mode 1 with two K-tiles in flight
// The source region per XCD is 2 MB (L2-resident), blocks start at different offsets.  Prints GB/s per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o mb_feed mb_feed.hip ; run: ./mb_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NW = 8;
constexpr int REGION = 2 * 1024 * 1024;      // bytes per XCD region
constexpr int ROW_STRIDE = 2560;             // bytes (K = 1280 fp16)

template <int MODE>
__global__ __launch_bounds__(512, 2) void feed_kernel(const char* __restrict__ src, int ktiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x & 7) * REGION;
    const int srow = lane >> 3, slot = lane & 7;
    unsigned off = (blockIdx.x >> 3) * 73728u;                    // staggered start inside the XCD's region
    f16x8 acc = {};
    constexpr int NV = (MODE == 4) ? 20 : ((MODE == 3) ? 5 : 9);
    f16x8 r[2][NV > 9 ? NV : 9];
    auto row_addr = [&](unsigned o, int piece) {                  // 8 rows x 128 B per wave instruction
        const unsigned rowbase = (o + (unsigned)(wave * 9 + piece) * 8u * ROW_STRIDE) % (REGION - 8 * ROW_STRIDE);
        return base + rowbase + srow * ROW_STRIDE + slot * 16;
    };
    auto lin_addr = [&](unsigned o, int piece, int w) {           // 1 KB contiguous per wave instruction
        const unsigned b = (o + (unsigned)(w * 9 + piece) * 1024u) % (REGION - 1024);
        return base + (b & ~15u) + lane * 16;
    };
    for (int v = 0; v < ktiles; ++v) {
        const int s = v & 1;
        char* buf = smem + s * 73728;
        if constexpr (MODE == 0 || MODE == 5) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                __builtin_amdgcn_global_load_lds((gptr_t)row_addr(off, k), (lptr_t)(buf + (wave * 9 + k) * 1024), 16, 0, 0);
            if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if constexpr (MODE == 1 || MODE == 2 || MODE == 6) {
            if constexpr (MODE == 6) {
                if (v > 0) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc += r[s ^ 1][k];           // consume the previous tile (forces its wait)
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k)
                r[MODE == 6 ? s : 0][k] = *(const f16x8*)(MODE == 2 ? row_addr(off, k) : lin_addr(off, k, wave));
            if constexpr (MODE != 6) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += r[0][k];
                __builtin_amdgcn_s_barrier();
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_global_load_lds((gptr_t)row_addr(off, k), (lptr_t)(buf + (wave * 4 + k) * 1024), 16, 0, 0);
#pragma unroll
            for (int k = 0; k < 5; ++k) r[0][k] = *(const f16x8*)lin_addr(off + 40000u, k, wave);
#pragma unroll
            for (int k = 0; k < 5; ++k) acc += r[0][k];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_global_load_lds((gptr_t)row_addr(off, k), (lptr_t)(buf + (wave * 4 + k) * 1024), 16, 0, 0);
            // every wave loads the 20 fragments of its N half (same addresses for the 4 M-waves of that half)
#pragma unroll
            for (int k = 0; k < 20; ++k) r[0][k] = *(const f16x8*)lin_addr(off + 40000u + (unsigned)k * 1024u, 0, wave & 1);
#pragma unroll
            for (int k = 0; k < 20; ++k) acc += r[0][k];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        off += 73728u;
        if (off >= (unsigned)REGION) off -= REGION;
    }
    if constexpr (MODE == 6) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += r[(ktiles - 1) & 1][k];
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t += (float)acc[e];
    if (t == 12345.678f) sink[tid] = t + smem[tid];
}

// ---- GEMM-like addressing: W = 320 rows (shared by every block), A = 256 rows (aliased to one row, or private per block); a
// K-tile is the 128-byte column block kt of those rows (row stride S bytes), kt advancing along the row like the real K loop.
template <bool ALIAS_A, int INFLIGHT>
__global__ __launch_bounds__(512, 2) void gemm_like_kernel(const char* __restrict__ wsrc, const char* __restrict__ asrc, int S,
                                                           int nk, int ktiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = lane >> 3, slot = (lane & 7) ^ srow;
    const char* ablk = asrc + (ALIAS_A ? 0 : (size_t)blockIdx.x * 256 * S);
    int kt = (blockIdx.x * 7) % nk;
    for (int v = 0; v < ktiles; ++v) {
        char* buf = smem + (v & 1) * 73728;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ALIAS_A ? 0 : (wave + 8 * i) * 8 + srow;
            __builtin_amdgcn_global_load_lds((gptr_t)(ablk + (size_t)row * S + kt * 128 + slot * 16), (lptr_t)(buf + (wave + 8 * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int row = (wave + 8 * j) * 8 + srow;
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)row * S + kt * 128 + slot * 16), (lptr_t)(buf + 32768 + (wave + 8 * j) * 1024), 16, 0, 0);
        }
        if constexpr (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (++kt == nk) kt = 0;
    }
    if (ktiles == -1) sink[tid] = smem[tid];
}


// ---- Round 4: the same GEMM-like feed with the A operand stored CHUNK-MAJOR, [K/64][M][64]: a 256-row K-tile is
// one contiguous 32 KB run (each wave instruction 1 KB contiguous) instead of 256 pieces of 128 B at pitch S.  SHARE = number of
// blocks of one XCD that read the same A rows (1: N = 320, A streamed once from HBM; 4: N = 1280, four tile columns in phase).
// WORK = dependent-FMA filler per K-tile (0: pure feed; ~1.9 us: the MFMA time of a 256x320x64 tile at 1400 TF/s), so the second
// regime measures whether the tile ARRIVES within one compute interval (latency), not the peak stream rate.
template <int LAYOUT, int SHARE, int INFLIGHT>
__global__ __launch_bounds__(512, 2) void gemm_like2_kernel(const char* __restrict__ wsrc, const char* __restrict__ asrc, int S,
                                                            int nk, int ktiles, int work, size_t mtot, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int srow = lane >> 3, slot = (lane & 7) ^ srow;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;              // 32 blocks per XCD
    const size_t mblk = (size_t)(xcd * (32 / SHARE) + idx / SHARE);     // A row block owned (shared by SHARE blocks of the XCD)
    int kt = (int)((mblk * 7) % nk);
    float f = (float)tid;
    for (int v = 0; v < ktiles; ++v) {
        char* buf = smem + (v & 1) * 73728;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave + 8 * i) * 8 + srow;
            const char* src = LAYOUT == 0 ? asrc + (mblk * 256 + row) * (size_t)S + kt * 128 + slot * 16
                                          : asrc + ((size_t)kt * mtot + mblk * 256 + row) * 128 + slot * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + (wave + 8 * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int row = (wave + 8 * j) * 8 + srow;
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)row * S + kt * 128 + slot * 16), (lptr_t)(buf + 32768 + (wave + 8 * j) * 1024), 16, 0, 0);
        }
        for (int w = 0; w < work; ++w) f = __builtin_fmaf(f, 1.0000001f, 0.5f);      // 4-clock dependent chain per iteration
        if constexpr (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (++kt == nk) kt = 0;
    }
    if (ktiles == -1 || f == 12345.678f) sink[tid] = smem[tid] + f;
}

template <int LAYOUT, int SHARE, int INFLIGHT>
void run_gemm_like2(const char* wsrc, const char* asrc, int S, int work, float* sink) {
    const int ktiles = 4000, nk = S / 128;
    const size_t lds = 2 * 73728;
    const size_t mtot = (size_t)256 * 256 / SHARE;
    CK(hipFuncSetAttribute((const void*)gemm_like2_kernel<LAYOUT, SHARE, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gemm_like2_kernel<LAYOUT, SHARE, INFLIGHT>), dim3(256), dim3(512), lds, 0, wsrc, asrc, S, nk, ktiles, work, mtot, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) printf("r4 S=%5d A %-11s share %d in-flight %d work %4d: %8.3f ms  %6.1f GB/s per CU staged, A from memory %5.2f TB/s, %6.3f us per K-tile\n", S,
                        LAYOUT ? "chunk-major" : "row-major", SHARE, INFLIGHT, work, ms, 73728.0 * ktiles / ms / 1e6,
                        32768.0 * ktiles * (256 / SHARE) / ms / 1e9, ms * 1e3 / ktiles);
    }
}

template <bool ALIAS_A, int INFLIGHT>
void run_gemm_like(const char* wsrc, const char* asrc, int S, float* sink, const char* label) {
    const int ktiles = 4000, nk = S / 128;
    const size_t lds = 2 * 73728;
    CK(hipFuncSetAttribute((const void*)gemm_like_kernel<ALIAS_A, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gemm_like_kernel<ALIAS_A, INFLIGHT>), dim3(256), dim3(512), lds, 0, wsrc, asrc, S, nk, ktiles, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) printf("gemm-like S=%5d %-12s in flight %d: %-40s %8.3f ms  %7.1f GB/s per CU\n", S, ALIAS_A ? "A aliased" : "A private", INFLIGHT, label, ms,
                        73728.0 * ktiles / ms / 1e6);
    }
}

template <int MODE>
void run(const char* src, float* sink, const char* label, double bytes_per_tile) {
    const int ktiles = 4000;
    const size_t lds = 2 * 73728;
    CK(hipFuncSetAttribute((const void*)feed_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(feed_kernel<MODE>, dim3(256), dim3(512), lds, 0, src, ktiles, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("mode %d %-62s %8.3f ms  %7.1f GB/s per CU  (%5.2f TB/s chip)\n", MODE, label, ms,
                        bytes_per_tile * ktiles / ms / 1e6, bytes_per_tile * ktiles * 256 / ms / 1e9);
    }
}

int main() {
    char* src; float* sink;
    CK(hipMalloc(&src, 8 * (size_t)REGION + 65536));
    CK(hipMalloc(&sink, 4096));
    std::vector<unsigned short> h(4 * (size_t)REGION);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    CK(hipMemcpy(src, h.data(), 8 * (size_t)REGION, hipMemcpyHostToDevice));
    run<0>(src, sink, "all LDS-DMA, 8 rows x 128 B per instr, 1 tile in flight", 73728);
    run<5>(src, sink, "all LDS-DMA, 2 tiles in flight", 73728);
    run<1>(src, sink, "all VGPR loads, 1 KB contiguous per instr, 1 tile in flight", 73728);
    run<6>(src, sink, "all VGPR loads contiguous, 2 tiles in flight", 73728);
    run<2>(src, sink, "all VGPR loads, 8 rows x 128 B per instr", 73728);
    run<3>(src, sink, "A 32 KB LDS-DMA + W 40 KB VGPR contiguous (1x per block)", 73728);
    run<4>(src, sink, "A 32 KB LDS-DMA + W VGPR, each wave its N half (4x redundant)", 32768 + 8 * 20480);
    char* big;
    const size_t big_bytes = (size_t)256 * 256 * 10496 + (size_t)320 * 10496 + 65536;
    CK(hipMalloc(&big, big_bytes));
    CK(hipMemset(big, 0x3c, big_bytes));
    for (int S : {2560, 5120, 10240, 10368}) {
        const char* w = big;
        const char* a = big + (size_t)320 * 10496;
        run_gemm_like<true, 1>(w, a, S, sink, "W 320 rows shared, A one line");
        run_gemm_like<true, 2>(w, a, S, sink, "W 320 rows shared, A one line");
        run_gemm_like<false, 1>(w, a, S, sink, "W shared, A 256 private rows per block");
        run_gemm_like<false, 2>(w, a, S, sink, "W shared, A 256 private rows per block");
    }
    // round 4: row-major vs chunk-major A.  work: 0 = pure feed; the filler loop costs ~14.5 ns per iteration (measured: 420 -> 6.2 us per
    // K-tile), so 60 / 120 / 180 = ~0.9 / 1.8 / 2.6 us of compute per K-tile (a 256x320x64 tile is ~1.9 us of MFMA time at 1400 TF/s)
    for (int S : {1280, 2560, 5120, 10240}) {
        const char* w = big;
        const char* a = big + (size_t)320 * 10496;
        for (int work : {0, 60, 120, 180}) {
            run_gemm_like2<0, 1, 1>(w, a, S, work, sink);
            run_gemm_like2<1, 1, 1>(w, a, S, work, sink);
            run_gemm_like2<0, 4, 1>(w, a, S, work, sink);
            run_gemm_like2<1, 4, 1>(w, a, S, work, sink);
        }
        run_gemm_like2<0, 1, 2>(w, a, S, 0, sink);
        run_gemm_like2<1, 1, 2>(w, a, S, 0, sink);
    }
    return 0;
}
