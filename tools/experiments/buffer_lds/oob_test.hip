// Does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose offset is outside num_records (gfx950)?  (round 6: the masked taps of the
// conv loader would then need no zero page and no 64-bit pointer select.)   hipcc --offload-arch=gfx950 -O3 oob_test.hip -o oob_test && ./oob_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned* src, unsigned* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0xdeadbeefu;                 // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // even lanes: in range (their own 16 bytes); odd lanes: far out of range
    const int voff = (lane & 1) ? (int)0x80000000 : lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 0x40000000);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const unsigned want = (l & 1) ? 0u : 0x1000 + l * 4 + e;
            if (r[l * 4 + e] != want) { ok = 0; if (l < 6) printf("lane %d elt %d: got %08x want %08x\n", l, e, r[l * 4 + e], want); }
        }
    printf(ok ? "OOB lanes wrote zeros, in-range lanes their data: OK\n" : "MISMATCH\n");
    return !ok;
}
