// geometry.hip -- Plücker embedding, depth lift and pano->perspective kernels (gfx950); the point-cloud / image kernels of
// the reprojection stage live in reproject.hip.  All HBM-bound:
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

}  // namespace

extern "C" ew_status ew_plucker_embed(const float* rays, const float* c2w, float* out, int N, int H, int W, void* stream) {
    EW_REQUIRE(rays && c2w && out && N > 0 && H > 0 && W > 0, "ew_plucker_embed: bad args");
    hipLaunchKernelGGL(plucker_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, rays, c2w,
                       out, N, H * W);
    return ew_check_launch("ew_plucker_embed");
}

extern "C" ew_status ew_depth_unproject(const float* depth, const float* extr, const float* intr, float* xyz, int S, int H,
                                        int W, void* stream) {
    EW_REQUIRE(depth && extr && intr && xyz && S > 0 && H > 0 && W > 0, "ew_depth_unproject: bad args");
    hipLaunchKernelGGL(depth_unproject_kernel, dim3(grid_for((long long)S * H * W)), dim3(256), 0, (hipStream_t)stream,
                       depth, extr, intr, xyz, S, H, W);
    return ew_check_launch("ew_depth_unproject");
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

