// mb_epi.hip -- stand-alone microbenchmark (round 4): what bounds the residual-carrying short-K GEMMs of level 0 at ~2.5 TB/s?
// (their time is proportional to bytes at half the rate the streaming norm kernels reach.)
// One persistent 8-wave workgroup per CU walks 256-row x 320-column tiles of a [M, 320] problem like gemm3 does, but only the MEMORY
// side: phase A = the A operand of a K = 320 GEMM through LDS-DMA (5 K-tiles of 256 rows x 128 B, one K-tile in flight), phase E =
// the epilogue's operand traffic: residual in (hi 2 B + lo8 1 B per element) and output (hi + lo8), in one of two lane mappings:
//   E0  the DIRECT epilogue's mapping: per wave instruction 16 rows x 64 B (lane = fks*16 + frow owns 8 consecutive columns of row frow)
//   E1  row-contiguous: per wave instruction 1.6 rows x 640 B (40 lanes x 16 B per row), the norm kernels' pattern
// with D = number of (row fragment) steps whose loads are requested before the first store (1 = shipped epilogue).
// Prints TB/s of the bytes moved.  Build: hipcc --offload-arch=gfx950 -O3 -o mb_epi mb_epi.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int C = 320, BM = 256, NW = 8;

// PH: bit 0 = phase A, bit 1 = phase E.  MAP: 0 = E0, 1 = E1.  LO: carry the lo8 planes too.
template <int PH, int MAP, bool LO>
__global__ __launch_bounds__(512, 2) void epi_kernel(const char* __restrict__ a, const char* __restrict__ r_hi, const char* __restrict__ r_lo,
                                                     char* __restrict__ o_hi, char* __restrict__ o_lo, int tiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fks = lane >> 4;
    float acc = 0.f;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t m0 = (size_t)t * BM;
        if (PH & 1) {
            // phase A: 5 K-tiles, each 256 rows x 128 B (row pitch 640 B): 32 pieces of 8 rows, 4 per wave; one K-tile in flight
            const int srow = lane >> 3, slot = lane & 7;
#pragma unroll 1
            for (int kt = 0; kt < 5; ++kt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const char* src = a + (m0 + (wave + NW * i) * 8 + srow) * (C * 2) + kt * 128 + slot * 16;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (kt & 1) * 32768 + (wave + NW * i) * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            acc += *(const float*)(smem + lane * 4);
        }
        if (PH & 2) {
            // phase E: the wave's 64 x 160 patch (wm, wn) of the tile
            if (MAP == 0) {
#pragma unroll 1
                for (int i = 0; i < 4; ++i) {
                    const size_t row = m0 + wm * 64 + i * 16 + frow;
                    f16x8 v[5];
                    u32x2 l[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const size_t e = row * C + wn * 160 + q * 32 + fks * 8;
                        v[q] = *(const f16x8*)(r_hi + e * 2);
                        if (LO) l[q] = *(const u32x2*)(r_lo + e);
                    }
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const size_t e = row * C + wn * 160 + q * 32 + fks * 8;
                        *(f16x8*)(o_hi + e * 2) = v[q] + v[q];
                        if (LO) *(u32x2*)(o_lo + e) = l[q] ^ 1u;
                    }
                }
            } else {
                // the same 64 x 160 patch as 64 rows x 20 vectors of 16 B: instruction k covers vectors [64k, 64k+64) in row-major order
#pragma unroll 1
                for (int i = 0; i < 4; ++i) {
                    f16x8 v[5];
                    u32x2 l[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const int idx = (i * 5 + q) * 64 + lane;
                        const size_t e = (m0 + wm * 64 + idx / 20) * C + wn * 160 + (idx % 20) * 8;
                        v[q] = *(const f16x8*)(r_hi + e * 2);
                        if (LO) l[q] = *(const u32x2*)(r_lo + e);
                    }
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const int idx = (i * 5 + q) * 64 + lane;
                        const size_t e = (m0 + wm * 64 + idx / 20) * C + wn * 160 + (idx % 20) * 8;
                        *(f16x8*)(o_hi + e * 2) = v[q] + v[q];
                        if (LO) *(u32x2*)(o_lo + e) = l[q] ^ 1u;
                    }
                }
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int PH, int MAP, bool LO>
void run(const char* a, const char* rh, const char* rl, char* oh, char* ol, int M, float* sink, const char* what) {
    const int tiles = M / BM;
    CK(hipFuncSetAttribute((const void*)epi_kernel<PH, MAP, LO>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t s, e;
    CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((epi_kernel<PH, MAP, LO>), dim3(256), dim3(512), 65536, 0, a, rh, rl, oh, ol, tiles, sink);
    CK(hipEventRecord(s));
    const int it = 10;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((epi_kernel<PH, MAP, LO>), dim3(256), dim3(512), 65536, 0, a, rh, rl, oh, ol, tiles, sink);
    CK(hipEventRecord(e));
    CK(hipEventSynchronize(e));
    float ms;
    CK(hipEventElapsedTime(&ms, s, e));
    ms /= it;
    const double el = (double)M * C;
    const double bytes = ((PH & 1) ? el * 2 : 0) + ((PH & 2) ? el * (LO ? 6 : 4) : 0);
    printf("%-78s %8.1f us  %6.1f MB  %5.2f TB/s\n", what, ms * 1e3, bytes / 1e6, bytes / ms / 1e9);
}

int main() {
    const int M = 460800;
    const size_t n = (size_t)M * C;
    char *a, *rh, *rl, *oh, *ol; float* sink;
    CK(hipMalloc(&a, n * 2)); CK(hipMalloc(&rh, n * 2)); CK(hipMalloc(&rl, n)); CK(hipMalloc(&oh, n * 2)); CK(hipMalloc(&ol, n)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0x3c, n * 2)); CK(hipMemset(rh, 0x3c, n * 2)); CK(hipMemset(rl, 1, n));
    printf("# [M = 460800, C = 320]: level-0 projection with a split residual; 256 persistent workgroups of 8 waves\n");
    run<1, 0, true>(a, rh, rl, oh, ol, M, sink, "A only (LDS-DMA, 5 K-tiles, 1 in flight)");
    run<2, 0, false>(a, rh, rl, oh, ol, M, sink, "E only, E0 (16 rows x 64 B per instruction), hi planes");
    run<2, 1, false>(a, rh, rl, oh, ol, M, sink, "E only, E1 (row-contiguous 640 B), hi planes");
    run<2, 0, true>(a, rh, rl, oh, ol, M, sink, "E only, E0, hi + lo8");
    run<2, 1, true>(a, rh, rl, oh, ol, M, sink, "E only, E1, hi + lo8");
    run<3, 0, true>(a, rh, rl, oh, ol, M, sink, "A then E (serial phases per tile), E0, hi + lo8   <- the shipped structure");
    run<3, 1, true>(a, rh, rl, oh, ol, M, sink, "A then E, E1, hi + lo8");
    run<3, 0, false>(a, rh, rl, oh, ol, M, sink, "A then E, E0, hi planes");
    run<3, 1, false>(a, rh, rl, oh, ol, M, sink, "A then E, E1, hi planes");
    return 0;
}
