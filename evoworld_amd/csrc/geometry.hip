// geometry.hip -- Plücker embedding and the reprojection stage kernels (gfx950).  All HBM-bound:
// coalesced loads/stores, no GEMM reshaping.  Compiled with -ffp-contract=off so that the integer index
// paths (splat pixel index, cube->equirect LUT gather) are bit-exact against the numpy oracle
// (oracle/reproject_ref.py), which evaluates the same fp32 expression trees without FMA contraction.
// Reference: utils/plucker_embedding.py:221-255; evoworld/reprojection/reproject_vggt_open3d_utils.py:542-666;
// unified_loop_consistency.py:299-334,352-367.
#include "common.h"

namespace {

inline int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b < 8192 ? (b > 0 ? b : 1) : 8192); }

// out[n, 0:3, p] = R_n d_p ; out[n, 3:6, p] = t_n x (R_n d_p)
__global__ void plucker_kernel(const float* __restrict__ rays, const float* __restrict__ c2w, float* __restrict__ out,
                               int N, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (long long)n * HW);
        const float* M = c2w + n * 12;
        const float dx = rays[p * 3 + 0], dy = rays[p * 3 + 1], dz = rays[p * 3 + 2];
        const float wx = M[0] * dx + M[1] * dy + M[2] * dz;
        const float wy = M[4] * dx + M[5] * dy + M[6] * dz;
        const float wz = M[8] * dx + M[9] * dy + M[10] * dz;
        const float tx = M[3], ty = M[7], tz = M[11];
        float* o = out + (long long)n * 6 * HW + p;
        o[0] = wx;
        o[(long long)HW] = wy;
        o[2LL * HW] = wz;
        o[3LL * HW] = ty * wz - tz * wy;
        o[4LL * HW] = tz * wx - tx * wz;
        o[5LL * HW] = tx * wy - ty * wx;
    }
}

// pano[v, r, c, :] = faces[v, lut.face, lut.v, lut.u, :]
__global__ void cube2equi_kernel(const uint8_t* __restrict__ faces, const int16_t* __restrict__ lut,
                                 uint8_t* __restrict__ pano, int V, int HW, int res) {
    const long long total = (long long)V * HW;
    const long long face_sz = (long long)res * res * 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / HW);
        const int p = (int)(i - (long long)v * HW);
        const int f = lut[p * 3 + 0], vv = lut[p * 3 + 1], uu = lut[p * 3 + 2];
        const uint8_t* src = faces + ((long long)v * 6 + f) * face_sz + ((long long)vv * res + uu) * 3;
        uint8_t* dst = pano + i * 3;
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
    }
}

// Xc = ((u-cx) z / fx, (v-cy) z / fy, z);  Xw = R^T (Xc - t)
__global__ void depth_unproject_kernel(const float* __restrict__ depth, const float* __restrict__ extr,
                                       const float* __restrict__ intr, float* __restrict__ xyz, int S, int H, int W) {
    const long long HW = (long long)H * W, total = (long long)S * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(i / HW);
        const int p = (int)(i - (long long)s * HW);
        const int v = p / W, u = p - v * W;
        const float* E = extr + s * 12;
        const float* K = intr + s * 9;
        const float z = depth[i];
        const float xc = ((float)u - K[2]) * z / K[0];
        const float yc = ((float)v - K[5]) * z / K[4];
        const float ax = xc - E[3], ay = yc - E[7], az = z - E[11];
        float* o = xyz + i * 3;
        o[0] = E[0] * ax + E[4] * ay + E[8] * az;
        o[1] = E[1] * ax + E[5] * ay + E[9] * az;
        o[2] = E[2] * ax + E[6] * ay + E[10] * az;
    }
}

// one thread per (point, view): try the 6 faces, nearest-pixel z-test with a 64-bit atomicMin
__global__ void splat_kernel(const float* __restrict__ xyz, long long npts, const float* __restrict__ w2c,
                             unsigned long long* __restrict__ zbuf, int V, int res, float fx, float fy, float cx, float cy,
                             float z_near) {
    const long long total = npts * V;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / npts);
        const long long p = i - (long long)v * npts;
        const float x = xyz[p * 3 + 0], y = xyz[p * 3 + 1], z = xyz[p * 3 + 2];
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const float* M = w2c + ((long long)v * 6 + f) * 12;
            const float zc = ((M[8] * x + M[9] * y) + M[10] * z) + M[11];
            if (!(zc > z_near)) continue;
            const float xc = ((M[0] * x + M[1] * y) + M[2] * z) + M[3];
            const float yc = ((M[4] * x + M[5] * y) + M[6] * z) + M[7];
            const float pu = (fx * xc) / zc + cx;
            const float pv = (fy * yc) / zc + cy;
            const float fu = floorf(pu), fv = floorf(pv);
            if (fu >= 0.f && fu < (float)res && fv >= 0.f && fv < (float)res) {
                const int iu = (int)fu, iv = (int)fv;
                const unsigned long long key = ((unsigned long long)__float_as_uint(zc) << 32) | (unsigned long long)(unsigned)p;
                atomicMin(zbuf + (((long long)v * 6 + f) * res + iv) * res + iu, key);
            }
        }
    }
}

__global__ void splat_resolve_kernel(const unsigned long long* __restrict__ zbuf, const uint8_t* __restrict__ rgb,
                                     uint8_t* __restrict__ faces, long long npix) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long key = zbuf[i];
        uint8_t r = 0, g = 0, b = 0;
        if (key != ~0ULL) {
            const unsigned idx = (unsigned)(key & 0xffffffffULL);
            r = rgb[(long long)idx * 3 + 0];
            g = rgb[(long long)idx * 3 + 1];
            b = rgb[(long long)idx * 3 + 2];
        }
        faces[i * 3 + 0] = r;
        faces[i * 3 + 1] = g;
        faces[i * 3 + 2] = b;
    }
}

// perspective pixel -> ray (RDF pinhole) -> rot -> equirect continuous index -> bilinear (wrap in x, clamp in y)
__global__ void equi2pers_kernel(const uint8_t* __restrict__ equi, const float* __restrict__ rot, uint8_t* __restrict__ out,
                                 int F, int He, int We, int Hp, int Wp, float focal) {
    const long long HWp = (long long)Hp * Wp, total = (long long)F * HWp;
    const float PI = 3.14159265358979323846f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i / HWp);
        const int p = (int)(i - (long long)f * HWp);
        const int py = p / Wp, px = p - py * Wp;
        const float* R = rot + f * 9;
        const float cxr = ((float)px - (float)Wp * 0.5f) / focal, cyr = ((float)py - (float)Hp * 0.5f) / focal;
        const float dx = R[0] * cxr + R[1] * cyr + R[2];
        const float dy = R[3] * cxr + R[4] * cyr + R[5];
        const float dz = R[6] * cxr + R[7] * cyr + R[8];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float lon = atan2f(dx, dz);
        const float lat = asinf(dy / nrm);
        float ui = lon * (float)We / (2.0f * PI) + (float)We * 0.5f + 0.5f;
        float uj = lat * (float)He / PI + (float)He * 0.5f + 0.5f;
        ui = ui - floorf(ui / (float)We) * (float)We;
        uj = fminf(fmaxf(uj, 0.f), (float)(He - 1));
        const float x0f = floorf(ui), y0f = floorf(uj);
        const float ax = ui - x0f, ay = uj - y0f;
        int x0 = (int)x0f, y0 = (int)y0f;
        x0 = x0 >= We ? x0 - We : x0;
        const int x1 = x0 + 1 >= We ? 0 : x0 + 1;
        const int y1 = y0 + 1 >= He ? He - 1 : y0 + 1;
        const uint8_t* base = equi + (long long)f * He * We * 3;
        uint8_t* o = out + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v00 = base[((long long)y0 * We + x0) * 3 + c], v01 = base[((long long)y0 * We + x1) * 3 + c];
            const float v10 = base[((long long)y1 * We + x0) * 3 + c], v11 = base[((long long)y1 * We + x1) * 3 + c];
            const float top = v00 * (1.f - ax) + v01 * ax, bot = v10 * (1.f - ax) + v11 * ax;
            const float val = top * (1.f - ay) + bot * ay;
            o[c] = (uint8_t)fminf(fmaxf(val, 0.f), 255.f);   // truncation, like numpy .astype(uint8)
        }
    }
}

// Pillow-exact antialiased resampling pass for 8-bit images (ImagingResampleHorizontal/Vertical_8bpc): fixed-point
// coefficients (22 fractional bits), out = clip8((2^21 + sum_k px[k]*kk[k]) >> 22).  One pass along `axis_stride`.
// src/dst: [V, n_lines, ...] u8 with 3 channels; pass over the resampled axis of length n_in -> n_out.
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
        s0 >>= 22; s1 >>= 22; s2 >>= 22;
        dp[0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
        dp[1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
        dp[2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
    }
}

// u8 HWC -> fp32 CHW in [-1,1]: (x / 255) * 2 - 1   (torchvision ToTensor + CustomRescale)
__global__ void u8_hwc_to_f32_chw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long long n_img, int HW) {
    const long long total = n_img * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / HW;
        const int p = (int)(i - img * HW);
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[(img * 3 + c) * HW + p] = ((float)src[i * 3 + c] / 255.0f) * 2.0f - 1.0f;
    }
}

// fp32 CHW in [-1,1] -> u8 HWC: round_half_even(clamp(x/2 + 0.5, 0, 1) * 255) -- what the pipeline's PIL output holds
// (diffusers VideoProcessor: (x/2+0.5).clamp(0,1) -> (.*255).round().astype(uint8); pipeline_evoworld.py:727-732)
__global__ void f32_chw_to_u8_hwc_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, long long n_img, int HW) {
    const long long total = n_img * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / HW;
        const int p = (int)(i - img * HW);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = src[(img * 3 + c) * HW + p] / 2.0f + 0.5f;
            dst[i * 3 + c] = (uint8_t)rintf(fminf(fmaxf(x, 0.f), 1.f) * 255.0f);
        }
    }
}

}  // namespace

extern "C" ew_status ew_f32_chw_to_u8_hwc(const float* src, uint8_t* dst, int V, int H, int W, void* stream) {
    EW_REQUIRE(src && dst && V > 0 && H > 0 && W > 0, "ew_f32_chw_to_u8_hwc: bad args");
    hipLaunchKernelGGL(f32_chw_to_u8_hwc_kernel, dim3(grid_for((long long)V * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, (long long)V, H * W);
    return ew_check_launch("ew_f32_chw_to_u8_hwc");
}

extern "C" ew_status ew_plucker_embed(const float* rays, const float* c2w, float* out, int N, int H, int W, void* stream) {
    EW_REQUIRE(rays && c2w && out && N > 0 && H > 0 && W > 0, "ew_plucker_embed: bad args");
    hipLaunchKernelGGL(plucker_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, rays, c2w,
                       out, N, H * W);
    return ew_check_launch("ew_plucker_embed");
}

extern "C" ew_status ew_cube2equi_gather(const uint8_t* faces, const int16_t* lut, uint8_t* pano, int V, int H, int W,
                                         int res, void* stream) {
    EW_REQUIRE(faces && lut && pano && V > 0 && H > 0 && W > 0 && res > 0, "ew_cube2equi_gather: bad args");
    hipLaunchKernelGGL(cube2equi_kernel, dim3(grid_for((long long)V * H * W)), dim3(256), 0, (hipStream_t)stream, faces,
                       lut, pano, V, H * W, res);
    return ew_check_launch("ew_cube2equi_gather");
}

extern "C" ew_status ew_depth_unproject(const float* depth, const float* extr, const float* intr, float* xyz, int S, int H,
                                        int W, void* stream) {
    EW_REQUIRE(depth && extr && intr && xyz && S > 0 && H > 0 && W > 0, "ew_depth_unproject: bad args");
    hipLaunchKernelGGL(depth_unproject_kernel, dim3(grid_for((long long)S * H * W)), dim3(256), 0, (hipStream_t)stream,
                       depth, extr, intr, xyz, S, H, W);
    return ew_check_launch("ew_depth_unproject");
}

extern "C" ew_status ew_splat_cubemap(const float* xyz, size_t npts, const float* w2c, unsigned long long* zbuf, int V,
                                      int res, float fx, float fy, float cx, float cy, float z_near, void* stream) {
    EW_REQUIRE(w2c && zbuf && V > 0 && res > 0, "ew_splat_cubemap: bad args");
    EW_REQUIRE(npts < 0xffffffffULL, "ew_splat_cubemap: npts must fit 32 bits");
    if (npts == 0) return EW_OK;   // empty cloud: z-buffers stay at their init value
    EW_REQUIRE(xyz, "ew_splat_cubemap: null xyz");
    hipLaunchKernelGGL(splat_kernel, dim3(grid_for((long long)npts * V)), dim3(256), 0, (hipStream_t)stream, xyz,
                       (long long)npts, w2c, zbuf, V, res, fx, fy, cx, cy, z_near);
    return ew_check_launch("ew_splat_cubemap");
}

extern "C" ew_status ew_splat_resolve(const unsigned long long* zbuf, const uint8_t* rgb, uint8_t* faces, int V, int res,
                                      void* stream) {
    EW_REQUIRE(zbuf && faces && V > 0 && res > 0, "ew_splat_resolve: bad args");
    const long long npix = (long long)V * 6 * res * res;
    hipLaunchKernelGGL(splat_resolve_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, zbuf, rgb, faces, npix);
    return ew_check_launch("ew_splat_resolve");
}

extern "C" ew_status ew_equi2pers(const uint8_t* equi, const float* rot, uint8_t* out, int F, int He, int We, int Hp, int Wp,
                                  float fov_x_deg, void* stream) {
    EW_REQUIRE(equi && rot && out && F > 0 && He > 0 && We > 0 && Hp > 0 && Wp > 0, "ew_equi2pers: bad args");
    EW_REQUIRE(fov_x_deg > 0.f && fov_x_deg < 180.f, "ew_equi2pers: fov_x must be in (0,180)");
    const float focal = (float)Wp / (2.0f * tanf(fov_x_deg * 3.14159265358979323846f / 360.0f));
    hipLaunchKernelGGL(equi2pers_kernel, dim3(grid_for((long long)F * Hp * Wp)), dim3(256), 0, (hipStream_t)stream, equi,
                       rot, out, F, He, We, Hp, Wp, focal);
    return ew_check_launch("ew_equi2pers");
}

extern "C" ew_status ew_resize_aa_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, const int* kk_h, const int* bounds_h,
                                     int ksize_h, const int* kk_v, const int* bounds_v, int ksize_v, int V, int Hi, int Wi,
                                     int Ho, int Wo, void* stream) {
    EW_REQUIRE(src && tmp && dst && kk_h && bounds_h && kk_v && bounds_v, "ew_resize_aa_u8: null pointer");
    EW_REQUIRE(V > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && ksize_h > 0 && ksize_v > 0, "ew_resize_aa_u8: bad shape");
    hipStream_t s = (hipStream_t)stream;
    // horizontal pass: [V,Hi,Wi,3] -> tmp [V,Hi,Wo,3]
    hipLaunchKernelGGL(resample_pass_kernel, dim3(grid_for((long long)V * Hi * Wo)), dim3(256), 0, s, src, tmp, kk_h, bounds_h,
                       ksize_h, (long long)V, Hi, Wi, Wo, (long long)Wi * 3, 3LL, (long long)Wo * 3, 3LL, (long long)Hi * Wi * 3,
                       (long long)Hi * Wo * 3);
    // vertical pass: lines = columns of tmp: [V,Hi,Wo,3] -> dst [V,Ho,Wo,3]
    hipLaunchKernelGGL(resample_pass_kernel, dim3(grid_for((long long)V * Wo * Ho)), dim3(256), 0, s, tmp, dst, kk_v, bounds_v,
                       ksize_v, (long long)V, Wo, Hi, Ho, 3LL, (long long)Wo * 3, 3LL, (long long)Wo * 3, (long long)Hi * Wo * 3,
                       (long long)Ho * Wo * 3);
    return ew_check_launch("ew_resize_aa_u8");
}

extern "C" ew_status ew_u8_hwc_to_f32_chw(const uint8_t* src, float* dst, int V, int H, int W, void* stream) {
    EW_REQUIRE(src && dst && V > 0 && H > 0 && W > 0, "ew_u8_hwc_to_f32_chw: bad args");
    hipLaunchKernelGGL(u8_hwc_to_f32_chw_kernel, dim3(grid_for((long long)V * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, (long long)V, H * W);
    return ew_check_launch("ew_u8_hwc_to_f32_chw");
}
