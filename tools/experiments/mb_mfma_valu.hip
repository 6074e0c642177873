// Round 3 microbenchmark: do MFMA and VALU issued by the waves of ONE SIMD overlap, and does it matter whether the MFMA accumulators
// live in arch VGPRs (what hipcc picks for every 256-register kernel of this library: "AGPRs: 0") or in AccVGPRs?
//   hipcc --offload-arch=gfx950 -O3 -o mb_mfma_valu mb_mfma_valu.hip && ./mb_mfma_valu
// Per loop iteration a wave issues NM MFMA 16x16x32 f16 (8 independent accumulators) and NV plain v_fma_f32 (independent registers), all
// through inline asm so that the two builds differ in the accumulator register class only.  Grid = 256 x (waves per SIMD x 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFMA_V(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(x, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
#define VEXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define VMAD64(d, a, b, c) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c) : "vcc")
#define VLSHLADD64(d, a, c) asm volatile("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(c))
#define VPK(x, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b))
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ACC_A, int NM, int NV, int KIND>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const unsigned laddr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    f32x4 ld[8] = {};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    float x[16];
    f32x2 y[8];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int i = 0; i < 8; ++i) y[i] = (f32x2){(float)i, (float)threadIdx.x};
    int ia[4];
    long long ip[4], ibase = (long long)(size_t)out;
    for (int i = 0; i < 4; ++i) { ia[i] = threadIdx.x * 3 + i; ip[i] = i; }
    const float c0 = 1.0001f, c1 = 0.5f;
    const f32x2 p0 = {1.0001f, 0.9999f}, p1 = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < NM) { if (ACC_A) MFMA_A(acc[g], a, b); else MFMA_V(acc[g], a, b); }
#pragma unroll
            for (int v = 0; v < NV / 8; ++v) {
                if (KIND == 3) { if (v == 0) { if (g == 0) DSR(ld[0], laddr, 0); if (g == 1) DSR(ld[1], laddr, 1024); if (g == 2) DSR(ld[2], laddr, 2048); if (g == 3) DSR(ld[3], laddr, 3072);
                                               if (g == 4) DSR(ld[4], laddr, 0); if (g == 5) DSR(ld[5], laddr, 1024); if (g == 6) DSR(ld[6], laddr, 2048); if (g == 7) DSR(ld[7], laddr, 3072); }
                                 if (v == 1) { if (g == 0) DSR(ld[4], laddr, 16384); if (g == 2) DSR(ld[5], laddr, 17408); if (g == 4) DSR(ld[6], laddr, 18432); if (g == 6) DSR(ld[7], laddr, 19456); } }
                else if (KIND == 4) { long long t; VMAD64(t, ia[g & 3], iters, ibase); VLSHLADD64(ip[g & 3], t, ibase); }
                else if (KIND == 5) { VLSHLADD64(ip[g & 3], ip[g & 3], ibase); }
                else if (KIND == 0) VFMA(x[(g * (NV / 8) + v) & 15], c0, c1);
                else if (KIND == 1) VEXP(x[(g * (NV / 8) + v) & 15]);
                else VPK(y[(g * (NV / 8) + v) & 7], p0, p1);
            }
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += y[i][0] + y[i][1];
    if (KIND == 3) for (int i = 0; i < 8; ++i) s += ld[i][0] + ld[i][2];
    if (KIND >= 4) for (int i = 0; i < 4; ++i) s += (float)ip[i];
    if (s == 12345.678f) out[0] = s;
}

template <int ACC_A, int NM, int NV, int KIND>
static float run(int threads, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ACC_A, NM, NV, KIND>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACC_A, NM, NV, KIND>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

#define ROW(NM, NV, KIND, label)                                                                                                   \
    for (int threads = 256; threads <= 512; threads += 256) {                                                                      \
        const float v = run<0, NM, NV, KIND>(threads, d, iters), a = run<1, NM, NV, KIND>(threads, d, iters);                      \
        printf("%-34s waves/SIMD %d: acc in VGPR %7.3f ms (%6.1f ns/iter)   acc in AGPR %7.3f ms (%6.1f ns/iter)\n", label,       \
               threads / 256, v, v * 1e6 / iters, a, a * 1e6 / iters);                                                           \
    }

int main() {
    float* d;
    hipMalloc(&d, 4);
    const int iters = 20000;
    ROW(8, 0, 0, "8 MFMA only");
    ROW(0, 32, 0, "32 v_fma_f32 only");
    ROW(8, 32, 0, "8 MFMA + 32 v_fma_f32");
    ROW(0, 64, 0, "64 v_fma_f32 only");
    ROW(8, 64, 0, "8 MFMA + 64 v_fma_f32");
    ROW(0, 16, 1, "16 v_exp_f32 only");
    ROW(8, 16, 1, "8 MFMA + 16 v_exp_f32");
    ROW(0, 8, 3, "8 ds_read_b128 only");
    ROW(8, 8, 3, "8 MFMA + 8 ds_read_b128");
    ROW(0, 16, 3, "12 ds_read_b128 only");
    ROW(8, 16, 3, "8 MFMA + 12 ds_read_b128");
    ROW(0, 8, 4, "8 x (v_mad_i64_i32 + v_lshl_add_u64) only");
    ROW(8, 8, 4, "8 MFMA + 8 x (mad_i64 + lshl_add_u64)");
    ROW(0, 8, 5, "8 v_lshl_add_u64 only");
    ROW(8, 8, 5, "8 MFMA + 8 v_lshl_add_u64");
    ROW(0, 32, 2, "32 v_pk_fma_f32 only");
    ROW(8, 32, 2, "8 MFMA + 32 v_pk_fma_f32");
    return 0;
}
