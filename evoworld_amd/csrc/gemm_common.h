// gemm_common.h -- parameter block shared by the GEMM kernel generations.
#pragma once
#include "common.h"

struct GemmP {
    const f16* a;
    const f16* a2;
    const f16* w;
    const f16* bias;
    const f16* rowbias;
    const f16* r1;
    const f16* r2;
    f16* out;
    const f16* zero_page;
    const int8_t* r1_lo;   // split residual stream: lo8 companions, one byte per element, same element strides (may be null)
    const int8_t* r2_lo;
    int8_t* out_lo;
    int M, N, K;
    int c1, c2, lda, lda2, ld_out, ld_r1, ld_r2, ld_rowbias;
    int mode, n_img, h_in, w_in, h_out, w_out, stride, upsample;
    int tB, tT, tP;
    int conv_shift;  // conv3x3: 0 = taps centred on oy*stride (padding 1), 1 = taps start at oy*stride (padding (0,1): diffusers Downsample2D(padding=0))
    int rows_per_group, act;
    float c_acc, c_r1, c_r2;
    int tiles_m, tiles_n;
    int band;  // generation 3: tile columns per band of the tile order (0 = plain tn-fastest order)
    int dbg;   // ew_set_gemm_debug: bit 2 (value 4) = generation 3 runs the whole-tile schedule (no stream-K tail / half split)
};

// generation 3, stream-K tail (gemm3_f16.hip), passed as a second kernel argument (GemmP is kept under 256 bytes: beyond that the
// by-value kernel argument was copied to scratch instead of being read from the kernarg segment): the last `tiles` (G <= tiles <
// 2G) output tiles are split along K over the G persistent blocks; 0 = off.  ws: uncached fp32 partial-accumulator slots, flags:
// one word per slot, epoch: the value a slot's flag takes when its partial of THIS launch is complete.
struct SkP {
    float* ws;
    unsigned* flags;
    unsigned epoch;
    int tiles, dp_rounds;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const f16* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

