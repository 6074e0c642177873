// reproject.hip -- the reprojection stage's point-cloud and image kernels (gfx950), second generation.
// HBM / L2-atomic bound integer and byte work: 16-byte accesses, every tensor read once where the algorithm allows it.
//   R2  percentile filter        ew_select_kth_f32 (radix select, 5 streaming passes, no sort) + ew_filter_compact
//                                (order-preserving stream compaction of xyz + colour, 3 kernels)
//   R4  point splat              ew_splat_cubemap: every point is read ONCE (16-byte loads, 4 points per thread) and tested
//                                against all views x 6 faces with the 3x4 matrices in SGPRs; a relaxed L2 read of the z-buffer
//                                cell screens out occluded fragments before the 64-bit atomicMin
//   R5  resolve                  ew_splat_resolve: 4 pixels per thread, dword stores
//   R6  cube -> equirect         ew_cube2equi_gather: 4 pixels per thread, the LUT is read once for all views
//   R7  Pillow-exact resize      ew_resize_aa_u8: horizontal pass from an LDS-staged row, vertical pass on dwords
// Compiled with -ffp-contract=off: the pixel-index arithmetic must round exactly like the numpy oracle.
// Reference: evoworld/reprojection/reproject_vggt_open3d_utils.py:174-222,294-310 (filter), :617-666 (render), :542-614
// (cube -> equirect); dataset/CameraTrajDataset.py:586-619 (resize).
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

inline int grid_for(long long n, int per_block = 256) {
    long long b = (n + per_block - 1) / per_block;
    return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384);
}

// ------------------------------------------------------------------------------------------------
// radix select: k-th and (k+1)-th smallest of n floats
// ------------------------------------------------------------------------------------------------
struct SelState {
    unsigned hist[256];
    unsigned prefix, rank, cnt_le, min_gt;
    float out[2];
};

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // monotone: key order == float order
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void sel_init_kernel(SelState* st, unsigned k) {
    const int t = threadIdx.x;
    st->hist[t] = 0;
    if (t == 0) { st->prefix = 0; st->rank = k; st->cnt_le = 0; st->min_gt = 0xffffffffu; }
}

// histogram of digit `pass` (8 bits, most significant first) over the elements whose higher digits equal the prefix
__global__ __launch_bounds__(256) void sel_hist_kernel(const float* __restrict__ x, size_t n, SelState* st, int pass) {
    __shared__ unsigned lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned mask_hi = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    const unsigned prefix = st->prefix;
    const size_t n4 = n / 4;
    const f32x4* x4 = (const f32x4*)x;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned key = f2key(v[e]);
            if ((key & mask_hi) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
        }
    }
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const unsigned key = f2key(x[i]);
            if ((key & mask_hi) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ void sel_pick_kernel(SelState* st, int pass) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = st->hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned rank = st->rank, cum = 0;
        int d = 0;
        for (; d < 255; ++d) {
            if (cum + h[d] > rank) break;
            cum += h[d];
        }
        st->prefix |= (unsigned)d << (24 - 8 * pass);
        st->rank = rank - cum;
    }
    st->hist[threadIdx.x] = 0;
}

// count of keys <= kth key, smallest key above it
__global__ __launch_bounds__(256) void sel_tail_kernel(const float* __restrict__ x, size_t n, SelState* st) {
    __shared__ unsigned s_cnt, s_min;
    if (threadIdx.x == 0) { s_cnt = 0; s_min = 0xffffffffu; }
    __syncthreads();
    const unsigned kk = st->prefix;
    unsigned cnt = 0, mn = 0xffffffffu;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned key = f2key(x[i]);
        cnt += key <= kk ? 1u : 0u;
        mn = key > kk ? min(mn, key) : mn;
    }
    atomicAdd(&s_cnt, cnt);
    atomicMin(&s_min, mn);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&st->cnt_le, s_cnt);
        atomicMin(&st->min_gt, s_min);
    }
}

__global__ void sel_final_kernel(SelState* st, unsigned k, float* out2) {
    const float a = key2f(st->prefix);
    const float b = (st->cnt_le > k + 1 || st->min_gt == 0xffffffffu) ? a : key2f(st->min_gt);
    out2[0] = a;
    out2[1] = b;
}

// ------------------------------------------------------------------------------------------------
// order-preserving compaction of the points with conf >= thr: xyz [n,3] f32 -> out_xyz, colour (x255, truncated) -> RGBX u32
// ------------------------------------------------------------------------------------------------
constexpr int CP_EPT = 8, CP_BLOCK = 256 * CP_EPT;

__global__ __launch_bounds__(256) void compact_count_kernel(const float* __restrict__ conf, size_t n, float thr,
                                                            unsigned* __restrict__ blk_cnt) {
    __shared__ unsigned s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * CP_BLOCK + (size_t)threadIdx.x * CP_EPT;
    unsigned c = 0;
#pragma unroll
    for (int e = 0; e < CP_EPT; ++e)
        if (base + e < n && conf[base + e] >= thr) ++c;
    c = (unsigned)wave_sum((float)c);           // <= 512 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0) atomicAdd(&s, c);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void compact_scan_kernel(unsigned* __restrict__ blk_cnt, unsigned nblk,
                                                            unsigned* __restrict__ total) {
    __shared__ unsigned buf[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (unsigned b0 = 0; b0 < nblk; b0 += 1024) {
        const unsigned i = b0 + threadIdx.x;
        const unsigned v = i < nblk ? blk_cnt[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                       // Hillis-Steele inclusive scan
            const unsigned t = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblk) blk_cnt[i] = carry + buf[threadIdx.x] - v;   // exclusive offset
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// img layout 0: [n,3] (NHWC flattened); 1: [S,3,hw] planes (NCHW), point i -> frame i / hw, pixel i % hw
__global__ __launch_bounds__(256) void compact_scatter_kernel(const float* __restrict__ conf, size_t n, float thr,
                                                              const float* __restrict__ xyz, const float* __restrict__ img,
                                                              int img_nchw, unsigned hw, const unsigned* __restrict__ blk_off,
                                                              float* __restrict__ out_xyz, unsigned* __restrict__ out_rgbx) {
    __shared__ unsigned wsum[4];
    const size_t base = (size_t)blockIdx.x * CP_BLOCK + (size_t)threadIdx.x * CP_EPT;
    unsigned keep = 0, c = 0;
#pragma unroll
    for (int e = 0; e < CP_EPT; ++e)
        if (base + e < n && conf[base + e] >= thr) { keep |= 1u << e; ++c; }
    // exclusive scan of c over the 256 threads: wave prefix by shuffles, then 4 wave totals through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned off = blk_off[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int e = 0; e < CP_EPT; ++e) {
        if (!(keep & (1u << e))) continue;
        const size_t i = base + e;
        out_xyz[(size_t)off * 3 + 0] = xyz[i * 3 + 0];
        out_xyz[(size_t)off * 3 + 1] = xyz[i * 3 + 1];
        out_xyz[(size_t)off * 3 + 2] = xyz[i * 3 + 2];
        float r, g, b;
        if (img_nchw) {
            const size_t f = i / hw, p = i - f * hw;
            r = img[(f * 3 + 0) * hw + p]; g = img[(f * 3 + 1) * hw + p]; b = img[(f * 3 + 2) * hw + p];
        } else {
            r = img[i * 3 + 0]; g = img[i * 3 + 1]; b = img[i * 3 + 2];
        }
        // (images * 255).astype(np.uint8): truncation (reproject_vggt_open3d_utils.py:286-292)
        const unsigned ur = (unsigned)(int)(r * 255.0f) & 255u, ug = (unsigned)(int)(g * 255.0f) & 255u, ub = (unsigned)(int)(b * 255.0f) & 255u;
        out_rgbx[off] = ur | (ug << 8) | (ub << 16);
        ++off;
    }
}

// ------------------------------------------------------------------------------------------------
// splat: 4 points per thread (three 16-byte loads), all views x faces per point; matrices are wave-uniform (SGPRs)
// ------------------------------------------------------------------------------------------------
// Every point is read once and walks all V*6 matrices (point-major).  A view-major variant (blockIdx.y = view, the cloud re-read per view so that
// all resident waves hit ONE view's 12.6 MB of z-buffers instead of all 302 MB) measured 2.83 vs 2.64 ms at 5 M points x 24 views: the kernel is
// bound by the rate of memory-side 8-byte transactions (~45 G fragments/s), not by the footprint -- removed in round 6.
__global__ __launch_bounds__(256) void splat_kernel(const float* __restrict__ xyz, unsigned npts, const float* __restrict__ w2c,
                                                    u64* __restrict__ zbuf, int V, int res, float fx, float fy, float cx,
                                                    float cy, float z_near) {
    const float fres = (float)res;
    const int vf0 = 0, vf1 = V * 6;
    for (size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; base < npts; base += (size_t)gridDim.x * 1024) {
        float px[4], py[4], pz[4];
        const int cnt = (int)min((size_t)4, (size_t)npts - base);
        if (cnt == 4) {
            const f32x4* s = (const f32x4*)(xyz + base * 3);
            const f32x4 a = s[0], b = s[1], c = s[2];
            px[0] = a[0]; py[0] = a[1]; pz[0] = a[2];
            px[1] = a[3]; py[1] = b[0]; pz[1] = b[1];
            px[2] = b[2]; py[2] = b[3]; pz[2] = c[0];
            px[3] = c[1]; py[3] = c[2]; pz[3] = c[3];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t p = min(base + q, (size_t)npts - 1);
                px[q] = xyz[p * 3 + 0]; py[q] = xyz[p * 3 + 1]; pz[q] = xyz[p * 3 + 2];
            }
        }
        for (int vf = vf0; vf < vf1; ++vf) {
            const float* M = w2c + (size_t)vf * 12;                 // uniform address: scalar loads
            const float m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5], m6 = M[6], m7 = M[7];
            const float m8 = M[8], m9 = M[9], m10 = M[10], m11 = M[11];
            u64* const zb = zbuf + (size_t)vf * res * res;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= cnt) break;
                const float x = px[q], y = py[q], z = pz[q];
                const float zc = ((m8 * x + m9 * y) + m10 * z) + m11;
                if (!(zc > z_near)) continue;
                const float xc = ((m0 * x + m1 * y) + m2 * z) + m3;
                const float yc = ((m4 * x + m5 * y) + m6 * z) + m7;
                const float pu = (fx * xc) / zc + cx;
                const float pv = (fy * yc) / zc + cy;
                const float fu = floorf(pu), fv = floorf(pv);
                if (fu >= 0.f && fu < fres && fv >= 0.f && fv < fres) {
                    const int iu = (int)fu, iv = (int)fv;
                    const u64 key = ((u64)__float_as_uint(zc) << 32) | (u64)(unsigned)(base + q);
                    u64* cell = zb + (size_t)iv * res + iu;
                    // the cell only ever decreases: a (possibly stale) read that is already <= key proves the fragment loses
                    const u64 cur = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (key < cur) atomicMin(cell, key);
                }
            }
        }
    }
}

// z-buffer init: every cell = 0xFFFF...F ("no fragment"), 16-byte stores
__global__ __launch_bounds__(256) void zfill_kernel(u64* __restrict__ zbuf, size_t ncell) {
    const u32x4 ones = {~0u, ~0u, ~0u, ~0u};
    const size_t n2 = ncell / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) *(u32x4*)(zbuf + i * 2) = ones;
    if ((ncell & 1) && blockIdx.x == 0 && threadIdx.x == 0) zbuf[ncell - 1] = ~0ull;
}

// z-buffer -> colours.  CI: colour stride in bytes per point (3: packed RGB, 4: RGBX words); CO: output channels (3 | 4)
template <int CI, int CO>
__global__ __launch_bounds__(256) void resolve_kernel(const u64* __restrict__ zbuf, const uint8_t* __restrict__ rgb,
                                                      uint8_t* __restrict__ faces, size_t npix) {
    const size_t nq = npix / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (size_t)gridDim.x * 256) {
        const u32x4 k01 = *(const u32x4*)(zbuf + i * 4), k23 = *(const u32x4*)(zbuf + i * 4 + 2);
        const unsigned lo[4] = {k01[0], k01[2], k23[0], k23[2]}, hi[4] = {k01[1], k01[3], k23[1], k23[3]};
        unsigned col[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            col[q] = 0;
            if (!(lo[q] == 0xffffffffu && hi[q] == 0xffffffffu)) {
                if constexpr (CI == 4) col[q] = ((const unsigned*)rgb)[lo[q]] & 0xffffffu;
                else {
                    const uint8_t* s = rgb + (size_t)lo[q] * 3;
                    col[q] = (unsigned)s[0] | ((unsigned)s[1] << 8) | ((unsigned)s[2] << 16);
                }
            }
        }
        if constexpr (CO == 4) {
            *(u32x4*)(faces + i * 16) = (u32x4){col[0], col[1], col[2], col[3]};
        } else {
            unsigned* d = (unsigned*)(faces + i * 12);
            d[0] = col[0] | (col[1] << 24);
            d[1] = (col[1] >> 8) | (col[2] << 16);
            d[2] = (col[2] >> 16) | (col[3] << 8);
        }
    }
    if (blockIdx.x == 0) {
        for (size_t i = nq * 4 + threadIdx.x; i < npix; i += 256) {
            const u64 key = zbuf[i];
            unsigned col = 0;
            if (key != ~0ULL) {
                const unsigned idx = (unsigned)key;
                if constexpr (CI == 4) col = ((const unsigned*)rgb)[idx] & 0xffffffu;
                else col = (unsigned)rgb[(size_t)idx * 3] | ((unsigned)rgb[(size_t)idx * 3 + 1] << 8) | ((unsigned)rgb[(size_t)idx * 3 + 2] << 16);
            }
            if constexpr (CO == 4) ((unsigned*)faces)[i] = col;
            else { faces[i * 3] = col & 255u; faces[i * 3 + 1] = (col >> 8) & 255u; faces[i * 3 + 2] = col >> 16; }
        }
    }
}

// pano[v, p, :] = faces[v, lut.face(p), lut.v(p), lut.u(p), :].  4 pixels per thread; the LUT entry is decoded ONCE and used
// for all V views.  CF: channels of `faces` (3 | 4); the panorama is always packed RGB.
template <int CF>
__global__ __launch_bounds__(256) void cube2equi_kernel(const uint8_t* __restrict__ faces, const int16_t* __restrict__ lut,
                                                        uint8_t* __restrict__ pano, int V, int HW, int res, int vec) {
    // vec: the 4-pixel path stores dwords at pano + (v*HW + 4i)*3, which is 4-byte aligned for every view only when
    // HW % 4 == 0 (and the base pointers are aligned); otherwise every pixel takes the byte path below
    const int nq = vec ? HW / 4 : 0;
    const size_t view_sz = (size_t)6 * res * res;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
        const u32x2* L = (const u32x2*)(lut + (size_t)i * 12);       // 12 int16 = 24 bytes, 8-byte aligned
        const u32x2 a = L[0], b = L[1], c = L[2];
        const unsigned w[6] = {a[0], a[1], b[0], b[1], c[0], c[1]};
        unsigned src[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            short e[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int idx = q * 3 + t;
                e[t] = (short)((w[idx >> 1] >> ((idx & 1) * 16)) & 0xffffu);
            }
            src[q] = ((unsigned)e[0] * res + (unsigned)e[1]) * res + (unsigned)e[2];
        }
        for (int v = 0; v < V; ++v) {
            const uint8_t* fb = faces + (size_t)v * view_sz * CF;
            unsigned col[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (CF == 4) col[q] = ((const unsigned*)fb)[src[q]] & 0xffffffu;
                else {
                    const uint8_t* s = fb + (size_t)src[q] * 3;
                    col[q] = (unsigned)s[0] | ((unsigned)s[1] << 8) | ((unsigned)s[2] << 16);
                }
            }
            unsigned* d = (unsigned*)(pano + ((size_t)v * HW + (size_t)i * 4) * 3);
            d[0] = col[0] | (col[1] << 24);
            d[1] = (col[1] >> 8) | (col[2] << 16);
            d[2] = (col[2] >> 16) | (col[3] << 8);
        }
    }
    {                                                                 // byte path: nothing when vec (HW % 4 == 0), every pixel otherwise
        for (int p = nq * 4 + blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
            const int f = lut[p * 3 + 0], vv = lut[p * 3 + 1], uu = lut[p * 3 + 2];
            const size_t s = ((size_t)f * res + vv) * res + uu;
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) pano[((size_t)v * HW + p) * 3 + ch] = faces[((size_t)v * view_sz + s) * CF + ch];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pillow-exact antialiased resize (ImagingResampleHorizontal/Vertical_8bpc): 22-bit fixed-point coefficients,
// out = clip8((2^21 + sum_k px[k]*kk[k]) >> 22)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned clip8(int s) { s >>= 22; return (unsigned)(s < 0 ? 0 : (s > 255 ? 255 : s)); }

// horizontal: one block per (image, row); the source row is staged in LDS with dword loads; 4 output pixels per thread
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         const int* __restrict__ kk, const int* __restrict__ bounds, int ksize,
                                                         int Wi, int Wo) {
    extern __shared__ unsigned row[];                                // Wi*3 bytes (Wi*3 % 4 == 0)
    const size_t line = blockIdx.x;
    const unsigned* s = (const unsigned*)(src + line * (size_t)Wi * 3);
    const int nd = Wi * 3 / 4;
    for (int i = threadIdx.x; i < nd; i += 256) row[i] = s[i];
    __syncthreads();
    const uint8_t* rb = (const uint8_t*)row;
    unsigned* d = (unsigned*)(dst + line * (size_t)Wo * 3);
    for (int x4 = threadIdx.x; x4 < Wo / 4; x4 += 256) {
        unsigned col[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xo = x4 * 4 + q;
            const int xmin = bounds[xo * 2], xcnt = bounds[xo * 2 + 1];
            const int* k = kk + (size_t)xo * ksize;
            const uint8_t* sp = rb + xmin * 3;
            int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
            for (int x = 0; x < xcnt; ++x) {
                const int c = k[x];
                s0 += sp[0] * c; s1 += sp[1] * c; s2 += sp[2] * c;
                sp += 3;
            }
            col[q] = clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16);
        }
        d[x4 * 3 + 0] = col[0] | (col[1] << 24);
        d[x4 * 3 + 1] = (col[1] >> 8) | (col[2] << 16);
        d[x4 * 3 + 2] = (col[2] >> 16) | (col[3] << 8);
    }
}

// vertical: rows are byte strings of length RB = Wo*3 (RB % 4 == 0); one dword (4 bytes of one output row) per thread
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         const int* __restrict__ kk, const int* __restrict__ bounds, int ksize,
                                                         long long n_img, int Hi, int Ho, int RB) {
    const int rd = RB / 4;
    const long long total = n_img * Ho * rd;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xd = (int)(i % rd);
        const long long t = i / rd;
        const int yo = (int)(t % Ho);
        const long long img = t / Ho;
        const int ymin = bounds[yo * 2], ycnt = bounds[yo * 2 + 1];
        const int* k = kk + (size_t)yo * ksize;
        const unsigned* sp = (const unsigned*)(src + (img * Hi + ymin) * (long long)RB) + xd;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21, s3 = 1 << 21;
        for (int y = 0; y < ycnt; ++y) {
            const unsigned w = *sp;
            const int c = k[y];
            s0 += (int)(w & 255u) * c; s1 += (int)((w >> 8) & 255u) * c; s2 += (int)((w >> 16) & 255u) * c; s3 += (int)(w >> 24) * c;
            sp += rd;
        }
        ((unsigned*)(dst + (img * Ho + yo) * (long long)RB))[xd] = clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16) | (clip8(s3) << 24);
    }
}

// generic pass (any size / alignment): one output pixel per thread, byte accesses
__global__ void resample_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ kk,
                                     const int* __restrict__ bounds, int ksize, long long n_img, int n_lines, int n_in,
                                     int n_out, long long src_line_stride, long long src_elem_stride,
                                     long long dst_line_stride, long long dst_elem_stride, long long src_img_stride,
                                     long long dst_img_stride) {
    const long long total = n_img * n_lines * n_out;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % n_out);
        const long long t = i / n_out;
        const int line = (int)(t % n_lines);
        const long long img = t / n_lines;
        const int xmin = bounds[xo * 2], xcnt = bounds[xo * 2 + 1];
        const int* k = kk + (long long)xo * ksize;
        const uint8_t* sp = src + img * src_img_stride + line * src_line_stride + xmin * src_elem_stride;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int x = 0; x < xcnt; ++x) {
            const int c = k[x];
            s0 += sp[0] * c; s1 += sp[1] * c; s2 += sp[2] * c;
            sp += src_elem_stride;
        }
        uint8_t* dp = dst + img * dst_img_stride + line * dst_line_stride + xo * dst_elem_stride;
        dp[0] = (uint8_t)clip8(s0);
        dp[1] = (uint8_t)clip8(s1);
        dp[2] = (uint8_t)clip8(s2);
    }
}

// u8 HWC <-> fp32 CHW, 4 pixels per thread (HW % 4 == 0), scalar tail otherwise
__global__ __launch_bounds__(256) void u8_hwc_to_f32_chw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                                long long n_img, int HW) {
    if (HW % 4 == 0) {
        const int hq = HW / 4;
        const long long total = n_img * hq;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long img = i / hq;
            const int p4 = (int)(i - img * hq);
            const unsigned* s = (const unsigned*)(src + (img * HW + (long long)p4 * 4) * 3);
            const unsigned w0 = s[0], w1 = s[1], w2 = s[2];
            const unsigned b[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 & 255u, (w1 >> 8) & 255u,
                                    (w1 >> 16) & 255u, w1 >> 24, w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = ((float)b[q * 3 + c] / 255.0f) * 2.0f - 1.0f;
                *(f32x4*)(dst + (img * 3 + c) * HW + (long long)p4 * 4) = o;
            }
        }
    } else {
        const long long total = n_img * HW;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long img = i / HW;
            const int p = (int)(i - img * HW);
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[(img * 3 + c) * HW + p] = ((float)src[i * 3 + c] / 255.0f) * 2.0f - 1.0f;
        }
    }
}

// fp32 CHW in [-1,1] -> u8 HWC: round_half_even(clamp(x/2 + 0.5, 0, 1) * 255) -- what the pipeline's PIL output holds
// (diffusers VideoProcessor: (x/2+0.5).clamp(0,1) -> (.*255).round().astype(uint8); pipeline_evoworld.py:727-732)
__device__ __forceinline__ unsigned quant8(float v) { return (unsigned)rintf(fminf(fmaxf(v / 2.0f + 0.5f, 0.f), 1.f) * 255.0f); }
__global__ __launch_bounds__(256) void f32_chw_to_u8_hwc_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst,
                                                                long long n_img, int HW) {
    if (HW % 4 == 0) {
        const int hq = HW / 4;
        const long long total = n_img * hq;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long img = i / hq;
            const int p4 = (int)(i - img * hq);
            unsigned b[12];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 v = *(const f32x4*)(src + (img * 3 + c) * HW + (long long)p4 * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q * 3 + c] = quant8(v[q]);
            }
            unsigned* d = (unsigned*)(dst + (img * HW + (long long)p4 * 4) * 3);
            d[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            d[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            d[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
        }
    } else {
        const long long total = n_img * HW;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long img = i / HW;
            const int p = (int)(i - img * HW);
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[i * 3 + c] = (uint8_t)quant8(src[(img * 3 + c) * HW + p]);
        }
    }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" size_t ew_select_workspace_bytes(void) { return sizeof(SelState); }

extern "C" ew_status ew_select_kth_f32(const float* x, size_t n, size_t k, void* ws, float* out2, void* stream) {
    EW_REQUIRE(x && ws && out2, "ew_select_kth_f32: null pointer");
    EW_REQUIRE(n > 0 && n < 0xffffffffULL && k < n, "ew_select_kth_f32: need 0 <= k < n < 2^32");
    EW_REQUIRE(((uintptr_t)x & 15) == 0, "ew_select_kth_f32: x must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    SelState* st = (SelState*)ws;
    hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(256), 0, s, st, (unsigned)k);
    const int grid = grid_for((long long)(n / 4 + 1), 256 * 4);
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(sel_hist_kernel, dim3(grid), dim3(256), 0, s, x, n, st, pass);
        hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(256), 0, s, st, pass);
    }
    hipLaunchKernelGGL(sel_tail_kernel, dim3(grid_for((long long)n, 256 * 16)), dim3(256), 0, s, x, n, st);
    hipLaunchKernelGGL(sel_final_kernel, dim3(1), dim3(1), 0, s, st, (unsigned)k, out2);
    return ew_check_launch("ew_select_kth_f32");
}

extern "C" size_t ew_filter_compact_workspace_bytes(size_t n) { return ((n + CP_BLOCK - 1) / CP_BLOCK + 1) * sizeof(unsigned); }

extern "C" ew_status ew_filter_compact(const float* conf, size_t n, float thr, const float* xyz, const float* img,
                                       int img_nchw, unsigned hw, float* out_xyz, unsigned* out_rgbx, void* ws,
                                       unsigned* total, void* stream) {
    EW_REQUIRE(conf && xyz && img && out_xyz && out_rgbx && ws && total, "ew_filter_compact: null pointer");
    EW_REQUIRE(n > 0 && n < 0xffffffffULL, "ew_filter_compact: need 0 < n < 2^32");
    EW_REQUIRE(!img_nchw || (hw > 0 && n % hw == 0), "ew_filter_compact: NCHW images need n %% hw == 0");
    hipStream_t s = (hipStream_t)stream;
    unsigned* blk = (unsigned*)ws;
    const unsigned nblk = (unsigned)((n + CP_BLOCK - 1) / CP_BLOCK);
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblk), dim3(256), 0, s, conf, n, thr, blk);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, blk, nblk, total);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nblk), dim3(256), 0, s, conf, n, thr, xyz, img, img_nchw, hw, blk, out_xyz,
                       out_rgbx);
    return ew_check_launch("ew_filter_compact");
}

extern "C" ew_status ew_splat_cubemap(const float* xyz, size_t npts, const float* w2c, unsigned long long* zbuf, int V,
                                      int res, float fx, float fy, float cx, float cy, float z_near, void* stream) {
    EW_REQUIRE(w2c && zbuf && V > 0 && res > 0, "ew_splat_cubemap: bad args");
    EW_REQUIRE(npts < 0xffffffffULL, "ew_splat_cubemap: npts must fit 32 bits");
    EW_REQUIRE(((uintptr_t)zbuf & 15) == 0, "ew_splat_cubemap: zbuf must be 16-byte aligned");
    // round 6: the z-buffers are initialised HERE (they were the caller's job -- a torch fill kernel on the product path -- until ABI 9)
    const size_t ncell = (size_t)V * 6 * res * res;
    hipLaunchKernelGGL(zfill_kernel, dim3(grid_for((long long)(ncell / 2 + 1), 256)), dim3(256), 0, (hipStream_t)stream, zbuf, ncell);
    if (npts == 0) return ew_check_launch("ew_splat_cubemap");   // empty cloud: every cell stays "no fragment"
    EW_REQUIRE(xyz && ((uintptr_t)xyz & 15) == 0, "ew_splat_cubemap: xyz must be non-null and 16-byte aligned");
    int grid = grid_for((long long)(npts + 3) / 4, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(splat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, xyz, (unsigned)npts, w2c, zbuf, V, res, fx, fy, cx, cy, z_near);
    return ew_check_launch("ew_splat_cubemap");
}

extern "C" ew_status ew_splat_resolve(const unsigned long long* zbuf, const uint8_t* rgb, int rgb_stride, uint8_t* faces,
                                      int face_channels, int V, int res, void* stream) {
    EW_REQUIRE(zbuf && faces && V > 0 && res > 0, "ew_splat_resolve: bad args");
    EW_REQUIRE((rgb_stride == 3 || rgb_stride == 4) && (face_channels == 3 || face_channels == 4),
               "ew_splat_resolve: rgb_stride and face_channels must be 3 or 4");
    const size_t npix = (size_t)V * 6 * res * res;
    const int grid = grid_for((long long)(npix / 4 + 1), 256);
    hipStream_t s = (hipStream_t)stream;
    if (rgb_stride == 4 && face_channels == 4) hipLaunchKernelGGL((resolve_kernel<4, 4>), dim3(grid), dim3(256), 0, s, zbuf, rgb, faces, npix);
    else if (rgb_stride == 4) hipLaunchKernelGGL((resolve_kernel<4, 3>), dim3(grid), dim3(256), 0, s, zbuf, rgb, faces, npix);
    else if (face_channels == 4) hipLaunchKernelGGL((resolve_kernel<3, 4>), dim3(grid), dim3(256), 0, s, zbuf, rgb, faces, npix);
    else hipLaunchKernelGGL((resolve_kernel<3, 3>), dim3(grid), dim3(256), 0, s, zbuf, rgb, faces, npix);
    return ew_check_launch("ew_splat_resolve");
}

extern "C" ew_status ew_cube2equi_gather(const uint8_t* faces, int face_channels, const int16_t* lut, uint8_t* pano, int V,
                                         int H, int W, int res, void* stream) {
    EW_REQUIRE(faces && lut && pano && V > 0 && H > 0 && W > 0 && res > 0, "ew_cube2equi_gather: bad args");
    EW_REQUIRE(face_channels == 3 || face_channels == 4, "ew_cube2equi_gather: face_channels must be 3 or 4");
    EW_REQUIRE(((uintptr_t)lut & 1) == 0, "ew_cube2equi_gather: lut must be 2-byte aligned");      // unaligned operands take the byte path
    EW_REQUIRE((long long)6 * res * res < (1LL << 31), "ew_cube2equi_gather: face too large");
    const int grid = grid_for((long long)H * W / 4 + 1, 256);
    hipStream_t s = (hipStream_t)stream;
    const int vec = ((H * W) % 4 == 0 && (((uintptr_t)pano | (uintptr_t)faces) & 3) == 0 && ((uintptr_t)lut & 7) == 0) ? 1 : 0;
    if (face_channels == 4) hipLaunchKernelGGL(cube2equi_kernel<4>, dim3(grid), dim3(256), 0, s, faces, lut, pano, V, H * W, res, vec);
    else hipLaunchKernelGGL(cube2equi_kernel<3>, dim3(grid), dim3(256), 0, s, faces, lut, pano, V, H * W, res, vec);
    return ew_check_launch("ew_cube2equi_gather");
}

extern "C" ew_status ew_resize_aa_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, const int* kk_h, const int* bounds_h,
                                     int ksize_h, const int* kk_v, const int* bounds_v, int ksize_v, int V, int Hi, int Wi,
                                     int Ho, int Wo, void* stream) {
    EW_REQUIRE(src && tmp && dst && kk_h && bounds_h && kk_v && bounds_v, "ew_resize_aa_u8: null pointer");
    EW_REQUIRE(V > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && ksize_h > 0 && ksize_v > 0, "ew_resize_aa_u8: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const bool al = (((uintptr_t)src | (uintptr_t)tmp | (uintptr_t)dst) & 3) == 0;
    // horizontal pass: [V,Hi,Wi,3] -> tmp [V,Hi,Wo,3]
    if (al && (Wi * 3) % 4 == 0 && Wo % 4 == 0 && (size_t)Wi * 3 <= 60 * 1024) {
        hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((size_t)V * Hi)), dim3(256), (size_t)Wi * 3, s, src, tmp, kk_h, bounds_h,
                           ksize_h, Wi, Wo);
    } else {
        hipLaunchKernelGGL(resample_pass_kernel, dim3(grid_for((long long)V * Hi * Wo)), dim3(256), 0, s, src, tmp, kk_h, bounds_h,
                           ksize_h, (long long)V, Hi, Wi, Wo, (long long)Wi * 3, 3LL, (long long)Wo * 3, 3LL,
                           (long long)Hi * Wi * 3, (long long)Hi * Wo * 3);
    }
    // vertical pass: tmp [V,Hi,Wo,3] -> dst [V,Ho,Wo,3]
    if (al && (Wo * 3) % 4 == 0) {
        hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for((long long)V * Ho * (Wo * 3 / 4))), dim3(256), 0, s, tmp, dst, kk_v,
                           bounds_v, ksize_v, (long long)V, Hi, Ho, Wo * 3);
    } else {
        hipLaunchKernelGGL(resample_pass_kernel, dim3(grid_for((long long)V * Wo * Ho)), dim3(256), 0, s, tmp, dst, kk_v, bounds_v,
                           ksize_v, (long long)V, Wo, Hi, Ho, 3LL, (long long)Wo * 3, 3LL, (long long)Wo * 3,
                           (long long)Hi * Wo * 3, (long long)Ho * Wo * 3);
    }
    return ew_check_launch("ew_resize_aa_u8");
}

extern "C" ew_status ew_u8_hwc_to_f32_chw(const uint8_t* src, float* dst, int V, int H, int W, void* stream) {
    EW_REQUIRE(src && dst && V > 0 && H > 0 && W > 0, "ew_u8_hwc_to_f32_chw: bad args");
    EW_REQUIRE((((uintptr_t)src & 3) | ((uintptr_t)dst & 15)) == 0, "ew_u8_hwc_to_f32_chw: alignment");
    hipLaunchKernelGGL(u8_hwc_to_f32_chw_kernel, dim3(grid_for((long long)V * H * W / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, (long long)V, H * W);
    return ew_check_launch("ew_u8_hwc_to_f32_chw");
}

extern "C" ew_status ew_f32_chw_to_u8_hwc(const float* src, uint8_t* dst, int V, int H, int W, void* stream) {
    EW_REQUIRE(src && dst && V > 0 && H > 0 && W > 0, "ew_f32_chw_to_u8_hwc: bad args");
    EW_REQUIRE((((uintptr_t)dst & 3) | ((uintptr_t)src & 15)) == 0, "ew_f32_chw_to_u8_hwc: alignment");
    hipLaunchKernelGGL(f32_chw_to_u8_hwc_kernel, dim3(grid_for((long long)V * H * W / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, (long long)V, H * W);
    return ew_check_launch("ew_f32_chw_to_u8_hwc");
}
