/* evoworld_hip.h -- C ABI of libevoworld_hip.so (gfx950 / MI355X).
 *
 * The reference (JiahaoPlus/EvoWorld) has no FFI / plugin registry: its hot path is Python calling
 * third-party libraries (diffusers / torch / open3d / pyequilib).  This header is the drop-in boundary
 * the Python call surface (evoworld_amd/ *.py, mirroring evoworld/pipeline, evoworld/trainer/unet_plucker,
 * utils/plucker_embedding, evoworld/reprojection) binds through ctypes.  Each entry point cites the
 * reference interface (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor's data_ptr()); the library
 *     never allocates, frees or synchronises; every call is asynchronous on `stream` (a hipStream_t
 *     passed as void*; NULL = the default stream).
 *   - activations are fp16 channels-last: [N, H, W, C] == [tokens, C] row-major.
 *   - return value: 0 = OK, <0 = error; ew_last_error() returns a thread-local message.
 */
#ifndef EVOWORLD_HIP_H
#define EVOWORLD_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int ew_status;
#define EW_OK 0
#define EW_ERR_INVALID_ARG (-1)
#define EW_ERR_UNSUPPORTED (-2)
#define EW_ERR_HIP (-3)

#define EW_ABI_VERSION 10
int ew_abi_version(void);
const char* ew_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Fused MFMA GEMM / implicit-GEMM convolution  (fp16 in, fp32 accumulate, fp16 out).
 *   acc[m][n] = sum_k A(m,k) * W[n][k]                    W row-major [N, K], K = taps*(c1+c2); in the conv modes the K axis
 *                                                         is ordered [channel chunk of 64][tap][64 channels] (chunk-major,
 *                                                         tap-minor: the taps of a chunk re-hit its input lines in L2)
 *   v   = acc + bias[n] + rowbias[(m / rows_per_group) * ld_rowbias + n]
 *   v   = act(v)            act 0: none; 1: SiLU; 2: GEGLU -> out has N/2 columns (see w layout note below); 3: erf GELU
 *   out = c_acc*v + c_r1*r1[m][n] + c_r2*r2[m][n]
 * A-operand addressing modes (the gather happens in the global->LDS DMA address, no im2col buffer):
 *   EW_A_DENSE   A(m, k)          = a[m*lda + k]                      (k < c1; then a2[m*lda2 + k-c1])
 *   EW_A_CONV3X3 m -> (img, oy, ox) on an [n_img, h_out, w_out] grid; tap = ky*3+kx;
 *                A(m, (tap, c))   = in[img, iy, ix, c], iy = oy*stride+ky-1, ix = ox*stride+kx-1, zero
 *                outside; with upsample=1 the input is read as nearest-x2 upsampled ([h_in,w_in] -> 2x).
 *   EW_A_CONVT3  m -> (b, t, p) on a [B, T, P] grid; A(m, (kt, c)) = in[b, t+kt-1, p, c], zero outside.
 * In the conv modes channels [0,c1) come from `a`, [c1,c1+c2) from `a2` (the up-block skip concat,
 * evoworld/trainer/unet_plucker.py:458-475 + diffusers up blocks) without materialising the concat.
 * Replaces: torch.nn.Linear / Conv2d / Conv3d(3,1,1) calls inside the diffusers blocks instantiated at
 * evoworld/trainer/unet_plucker.py:126-244 (ResnetBlock2D, TemporalResnetBlock, Down/Upsample2D, Attention
 * projections, FeedForward/GEGLU, proj_in/out) and the AlphaBlender / residual adds around them.
 * Constraints: c1 % 64 == 0, c2 % 64 == 0 (pad channels), all pointers 16-byte aligned, ld* % 8 == 0.
 * GEGLU weight layout: rows are pre-interleaved in blocks of 16: tile rows [32q,32q+16) = value rows
 * 16q.., [32q+16,32q+32) = gate rows (N/2 + 16q..); bias likewise (evoworld_amd.unet packs this).
 */
enum { EW_A_DENSE = 0, EW_A_CONV3X3 = 1, EW_A_CONVT3 = 2 };
enum { EW_ACT_NONE = 0, EW_ACT_SILU = 1, EW_ACT_GEGLU = 2, EW_ACT_GELU = 3 /* erf GELU (CLIP ViT-H MLP) */ };

typedef struct ew_gemm_args {
    const void* a;      /* fp16 */
    const void* a2;     /* fp16 or NULL */
    const void* w;      /* fp16 [N, K] */
    const void* bias;   /* fp16 [N] or NULL */
    const void* rowbias;/* fp16 [G, N] or NULL */
    const void* r1;     /* fp16 [M, ld_r1] or NULL */
    const void* r2;     /* fp16 [M, ld_r2] or NULL */
    void* out;          /* fp16 [M, ld_out] */
    const void* zero_page; /* >= 256 bytes of device zeros (conv padding source) */
    int M, N;
    int c1, c2;         /* channels from a / a2 per tap */
    int lda, lda2;      /* row strides (elements) of a / a2 */
    int ld_out, ld_r1, ld_r2, ld_rowbias;
    int mode;           /* EW_A_* */
    int n_img, h_in, w_in, h_out, w_out, stride, upsample; /* EW_A_CONV3X3 */
    int tB, tT, tP;     /* EW_A_CONVT3 */
    int rows_per_group; /* rowbias group size in rows (>=1) */
    int act;            /* EW_ACT_* */
    float c_acc, c_r1, c_r2;
    /* Split residual stream (ABI 4): a residual-stream tensor x is carried as hi = fp16(x) (round to nearest even) plus an
     * int8 companion lo8, ONE byte per element with the same element strides: bits(x) ~= bits((float)hi) + 32 * lo8 on the
     * fp32 bit patterns, i.e. 8 more mantissa bits (~19 in all; the reference keeps the stream in fp32,
     * unified_loop_consistency.py:188) for 3 bytes per element.  Consumers that only need an fp16 operand read `hi` alone.
     * r1_lo / r2_lo (element strides ld_r1 / ld_r2) refine r1 / r2; out_lo (element stride ld_out) receives the lo8 of the
     * fp32 result.  Any of them may be NULL; not available with GEGLU.  (ABI 2-3 carried a second fp16 plane instead.) */
    const void* r1_lo;
    const void* r2_lo;
    void* out_lo;
    /* EW_A_CONV3X3 tap origin: 0 = iy = oy*stride + ky - 1 (symmetric padding 1); 1 = iy = oy*stride + ky, zero beyond the
     * bottom / right edge (F.pad(x, (0,1,0,1)) + stride-2 conv of diffusers Downsample2D(padding=0), the VAE encoder). */
    int conv_shift;
} ew_gemm_args;

ew_status ew_gemm_f16(const ew_gemm_args* args, void* stream);
/* Kernel generation behind ew_gemm_f16: 1 = 128x160 tile, 2 blocks/CU; 2 = persistent 3-stage ring, 256x160 / 128x256 tiles;
 * 3 (default) = persistent 256x320 (and 256x256) tile with a stream-K tail where N % 320 == 0 (or
 * N % 256 == 0) and the problem fills the chip, generation 2 for everything else.  Same arguments, same results to rounding;
 * kept selectable for A/B measurements. */
void ew_set_gemm_generation(int gen);
int ew_get_gemm_generation(void);
/* Debug aid for measurement tools (bench.py): rocprof-style name of the kernel variant the last ew_gemm_f16 call on this
   thread's library instance launched, e.g. "gemm3_kernel<0, 8>".  Not part of the reference surface. */
const char* ew_gemm_last_kernel(void);
void ew_set_gemm_debug(int flags);   /* bit 2 (value 4): generation 3 runs the whole-tile schedule -- no stream-K tail, no half split -- the twin the stream-K parity
                                        test compares against (same results up to fp32 summation order); 0 = normal.  (ABI <= 9 also had result-destroying
                                        ablation bits 0 / 1 and the halo-slab loader bits 3 / 4: removed in ABI 10.) */
/* Generation 3 splits the last round of output tiles along K over its 256 persistent workgroups when whole-tile rounds would
 * leave > 4 % of the chip idle (stream-K tail: fp32 partial accumulators handed over through a library-owned uncached
 * workspace, one per (device, stream), allocated on first use: 84 MB + 67 MB for the 256-wide instance).  Deterministic: the
 * split depends on the shape only.  A consumed hand-over flag is cleared by its consumer, so a captured launch can be replayed.
 * ew_gemm_streamk_status() synchronises and returns 0 when every hand-over completed, 1 if a finisher ever timed out
 * (results of that launch are invalid), -1 on a HIP error.  EW_G3_SK=0 in the environment switches the split off. */
int ew_gemm_streamk_status(void);
/* Allocates the stream-K workspace of (current device, stream).  Optional in eager use (the first launch that wants the tail
 * allocates it, under a mutex); call it BEFORE capturing launches into a hipGraph -- allocation is illegal during capture, and a
 * captured launch on a stream without a workspace runs the whole-tile schedule instead. */
ew_status ew_gemm_streamk_init(void* stream);

/* CU budget of the persistent kernels (ABI 8).  The persistent GEMM / feed-forward kernels launch one workgroup per CU and size their
 * tile schedule and stream-K split from the CU count: 256 on an MI355X.  A caller that runs the library on a stream restricted to a
 * subset of the CUs (ew_stream_create_cu_mask: two independent forwards -- the two rows of the CFG batch the reference concatenates,
 * evoworld/pipeline/pipeline_evoworld.py:689-711 -- side by side on disjoint halves of the chip) sets the budget to that subset's size
 * first: with more workgroups than CUs a stream-K finisher could wait for a contributor that is not resident.  Process-wide; a multiple
 * of 8 in [8, 256].  Returns the previous value.  Set it BEFORE any stream of the partition is in use: the value is atomic, but a launch reads it
 * more than once, so a change racing a launch on another thread may size that launch inconsistently. */
int ew_set_cu_budget(int n_cus);
int ew_get_cu_budget(void);
/* A HIP stream whose kernels only run on CUs [first_cu, first_cu + n_cus) of the CU-mask bit order (hipExtStreamCreateWithCUMask).
 * Returns NULL and sets ew_last_error on failure.  Destroy with ew_stream_destroy. */
void* ew_stream_create_cu_mask(int first_cu, int n_cus);
ew_status ew_stream_destroy(void* stream);

/* Fused GEGLU feed-forward for 320-channel tokens (ABI 5; level 0 of the U-Net: 460800 tokens per forward, 15 feed-forwards):
 *   out = c_acc * (GEGLU(x W1^T + b1) W2^T + b2 + rowbias[m / rows_per_group]) + c_r1 * r1 + c_r2 * r2
 * = diffusers FeedForward(320, activation_fn="geglu") (net.0 GEGLU projection 320 -> 2 x 1280, net.2 Linear 1280 -> 320) of
 * BasicTransformerBlock.ff / TemporalBasicTransformerBlock.ff_in / .ff (instantiated via evoworld/trainer/unet_plucker.py:161-233)
 * with the residual / AlphaBlender epilogue of ew_gemm_f16; the 1280-wide intermediate never goes to HBM.  x: fp16 [M, 320]
 * (the LayerNorm output); w1p / b1p / w2p: the weights in the kernel's LDS-image packs (layout: csrc/ff_fused.hip, built by
 * evoworld_amd.ops.ff_pack); b2 fp16 [320]; r1 / r2 / out [M, 320] with optional lo8 companions as in ew_gemm_args. */
typedef struct ew_ff_args {
    const void* x;
    const void* w1p;
    const void* b1p;
    const void* w2p;
    const void* b2;
    const void* rowbias;   /* fp16 [G, ld_rowbias] or NULL */
    const void* r1;
    const void* r1_lo;
    const void* r2;
    const void* r2_lo;
    void* out;
    void* out_lo;
    const void* zero_page;
    int M, C, hidden;      /* C = 320, hidden = 1280 */
    int rows_per_group, ld_rowbias;
    float c_acc, c_r1, c_r2;
    /* (ABI 5-9 carried an optional LayerNorm prologue here -- x_lo / ln_gamma / ln_beta / addvec / add_rows_per_group / ln_eps / ln_folded; both
     * forms measured slower than the separate ew_layernorm_f16 launch on the final kernels (+9 ms and +1.1 ms per forward)
     * and were removed in ABI 10.) */
} ew_ff_args;
ew_status ew_ff_geglu320_f16(const ew_ff_args* args, void* stream);

/* GroupNorm statistics + apply, channels-last fp16, over a (virtual) channel concat.
 * The normalised tensor has C_tot channels in `groups` groups; a stats/apply call handles the C_src channels
 * [c_off, c_off+C_src) that live in tensor `x` ([n_slabs*rows, C_src]); a skip-concat input is covered by
 * two calls (one per source) -- groups may straddle the seam.  x_lo (may be NULL) is the lo8 companion of a
 * split residual stream (int8, one byte per element, see ew_gemm_args).
 * Statistics are DETERMINISTIC and cancellation-safe (no atomics): ew_groupnorm_stats_f16 writes per-(slab, row chunk,
 * channel) sums of (x - K_c) and (x - K_c)^2 shifted by the pivot K_c = x[slab, row 0, c] into the workspace;
 * ew_groupnorm_finalize (one call per GroupNorm, after the stats calls of all its sources) reduces them in a fixed
 * order to (mean, biased variance) per (slab, group); ew_groupnorm_apply_f16 reads those.
 * `ws`: ew_groupnorm_workspace_floats(n_slabs, rows, C_tot, groups) floats, uninitialised, private to this GroupNorm call.
 *        slab n = `rows` consecutive rows (rows = H*W for the per-frame GN of ResnetBlock2D / transformer
 *        norm / conv_norm_out; rows = T*H*W for TemporalResnetBlock's GN over [B,C,T,H,W]).
 * apply: y[row][c_off+c] = (x-mean)*rstd*gamma[c_off+c]+beta[c_off+c], optional SiLU; y row stride = C_tot.
 * Replaces torch.nn.GroupNorm + SiLU at the diffusers blocks instantiated by
 * evoworld/trainer/unet_plucker.py:161-233 and conv_norm_out (:236, 478-479). */
size_t ew_groupnorm_workspace_floats(int n_slabs, int rows, int C_tot, int groups);
ew_status ew_groupnorm_stats_f16(const void* x, const void* x_lo, float* ws, int n_slabs, int rows, int C_src, int c_off,
                                 int C_tot, int groups, void* stream);
ew_status ew_groupnorm_finalize(float* ws, int n_slabs, int rows, int C_tot, int groups, void* stream);
ew_status ew_groupnorm_apply_f16(const void* x, const void* x_lo, const float* ws, const void* gamma, const void* beta,
                                 void* y, int n_slabs, int rows, int C_src, int c_off, int C_tot, int groups, float eps,
                                 int silu, void* stream);
/* Split-operand form of the apply (ABI 10): with f the fp32 result, y = fp16(f) and y_lo = fp16(f - float(y)), both with row stride ld_y
 * (>= C_tot).  y_lo = y + C_tot with ld_y = 2 * C_tot gives rows [x_hi | x_lo]: the A operand of a consumer whose weights are packed
 * [W_hi | W_hi | W_lo] over three K blocks (second source a2 = a, c2 = C_tot), i.e. x W formed to ~2^-21 in both operands.  Used where the
 * per-group energy analysis (tests/analysis_fp16_floor.py --per-group) puts a large share of the fp16 operand-rounding distance to the
 * reference's fp32 result in layers that are < 1 % of the flops: conv_norm_out -> conv_out (unet_plucker.py:236, 478-480) and the
 * level-0 TransformerSpatioTemporalModel.norm -> proj_in. */
ew_status ew_groupnorm_apply_split_f16(const void* x, const void* x_lo, const float* ws, const void* gamma, const void* beta,
                                       void* y, void* y_lo, int ld_y, int n_slabs, int rows, int C_src, int c_off, int C_tot,
                                       int groups, float eps, int silu, void* stream);

/* LayerNorm over the last dim (fp16 in/out, fp32 two-pass statistics).  x_lo (may be NULL): lo8 companion of a split
 * residual stream (int8).  Optional fused pre-add: x' = x + addvec[row / rows_per_group][:] is what gets normalised, and x' is
 * written to x_out (fp16) -- plus its lo8 companion to x_out_lo when non-NULL -- (the time_pos_embed add in
 * TransformerSpatioTemporalModel).  C % 8 == 0, C <= 2048.
 * Replaces torch.nn.LayerNorm in Basic/TemporalBasicTransformerBlock (diffusers, via unet_plucker.py:13). */
ew_status ew_layernorm_f16(const void* x, const void* x_lo, const void* addvec, int rows_per_group, void* x_out,
                           void* x_out_lo, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                           void* stream);

/* Spatial self-attention core, head_dim 64: o = softmax(q k^T * scale) v per (sequence, head), flash-tiled
 * on MFMA 32x32x16 with the swapped product S^T = K Q^T so each query's softmax row is lane-local.
 * q,k: fp16 token-major with row stride ld_qk (head h at +64h); vt: V TRANSPOSED [heads*64, n_seq*S]
 * (row = channel, tokens contiguous; produced by ew_gemm_f16 with swapped operands); o: [n_seq*S, ld_o].
 * Replaces F.scaled_dot_product_attention in AttnProcessor2_0 for BasicTransformerBlock.attn1
 * (SURVEY.md §8a U10). */
ew_status ew_attn_spatial_f16(const void* q, const void* k, const void* vt, void* o, int n_seq, int S, int heads,
                              int ld_qk, long long ld_vt, int ld_o, float scale, void* stream);

/* Same core for q and k that arrive PRE-SCALED (ABI 6): the projection GEMM's epilogue multiplied both by sqrt(scale * log2 e)
 * (ew_gemm_args.c_acc, applied in fp32 before the single rounding to fp16, so q and k carry the same relative rounding error as
 * unscaled ones), which makes q.k the softmax exponent in log2 units.  The kernel then lets the MFMA subtract the running maximum
 * (C operand = -m) and drops the per-score multiply-add: o = softmax_2(q k^T) v.  Same layouts and restrictions as above.
 * Replaces the same F.scaled_dot_product_attention call. */
ew_status ew_attn_spatial_log2_f16(const void* q, const void* k, const void* vt, void* o, int n_seq, int S, int heads,
                                   int ld_qk, long long ld_vt, int ld_o, void* stream);

/* Temporal self-attention core over the frame axis (T <= 64, head_dim 64; T <= 32 runs the one-block kernel) on frame-major tokens
 * [B, T, S, *]: sequence (b, s) attends over t -- the [B*T,S,C] <-> [B*S,T,C] regroup of
 * TemporalBasicTransformerBlock is done by addressing, not by a copy.  q,k,v row stride ld; o row stride ld_o.
 * Replaces SDPA in TemporalBasicTransformerBlock.attn1. */
ew_status ew_attn_temporal_f16(const void* q, const void* k, const void* v, void* o, int B, int T, int S, int heads,
                               int ld, int ld_o, float scale, void* stream);

/* Layout changes at the U-Net boundary.
 * nchw->nhwc: y[n,h,w,c_off+c] = fp16(scale * x[n,c,h,w]) for c < C (row stride ldc; other channels untouched).
 * nhwc->nchw: y[n,c,h,w] = fp32(x[n,h,w,c]) for c < C. */
ew_status ew_nchw_f32_to_nhwc_f16(const float* x, void* y, int N, int C, int H, int W, int ldc, int c_off, float scale,
                                  void* stream);
ew_status ew_nhwc_f16_to_nchw_f32(const void* x, float* y, int N, int C, int H, int W, int ldc, void* stream);
/* Split-operand form (ABI 9): with v = scale * x[n,c,h,w] and hi = fp16(v) the row receives THREE channel blocks,
 *   y[.., c_off + c] = hi,   y[.., lo_off + c_off + c] = fp16(v - hi),   y[.., dup_off + c_off + c] = fp16(hi * 2^-10),
 * i.e. the A operand [x_hi | x_lo | x_hi 2^-10] of a conv_in whose weight rows are packed [W_hi | W_hi | W_lo 2^10] (the power of two keeps
 * W_lo ~ 2^-12 |W| a normal fp16 number): x W is then formed to ~2^-21
 * instead of 2^-11 in BOTH operands at no MFMA cost, because conv_in pads its 18 input channels to one 64-channel K tile anyway
 * (3 * 20 <= 64).  The reference's conv_in runs on fp32 operands (weight_dtype = torch.float32, unified_loop_consistency.py:188;
 * evoworld/trainer/unet_plucker.py:131-136); its two fp16 roundings were 12-15 % each of the build's squared distance to the fp32 oracle. */
ew_status ew_nchw_f32_to_nhwc_split_f16(const float* x, void* y, int N, int C, int H, int W, int ldc, int c_off, int lo_off,
                                        int dup_off, float scale, void* stream);

/* Sinusoidal embedding of scalars (ABI 10): out fp16 [n_rows, dim] = [cos(v f_j) | sin(v f_j)], j < dim / 2, f_j = exp(-ln(10000) j / (dim / 2)),
 * v = vals[row % n_vals] (device fp32).  diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)` as the reference's U-Net applies it
 * to the timestep and to the three added time ids in every forward (evoworld/trainer/unet_plucker.py:395-420): two launches per forward instead of
 * the arange / exp / mul / cos / sin / cat chain of torch elementwise kernels. */
ew_status ew_sinusoid_embed_f16(const float* vals, int n_vals, int n_rows, int dim, void* out, void* stream);

/* Fused denoise-step glue: CFG combine + Euler (v-prediction) step + scale_model_input + concat for the next
 * step.  eps: fp16 NHWC [2*T, h, w, ld_eps] (rows [0,T) uncond, [T,2T) cond; first 4 channels);
 * latents: fp32 [T,4,h,w] state (updated in place); guidance [T]; next_in: fp16 NHWC [2*T,h,w,cpad], channels
 * [0,4) are rewritten with latents_new / sqrt(sigma_next^2+1) for both CFG rows (cond channels untouched).
 * Replaces evoworld/pipeline/pipeline_evoworld.py:691-695,709-714 + EulerDiscreteScheduler.step/scale_model_input. */
ew_status ew_euler_cfg_step(const void* eps, int ld_eps, float* latents, const float* guidance, float sigma,
                            float sigma_next, void* next_in, int cpad, int T, int h, int w, void* stream);
/* Same step writing the next model input as a split operand (ABI 9; see ew_nchw_f32_to_nhwc_split_f16): channels [0,4) = hi,
 * [lo_off, lo_off+4) = fp16(value - hi), [dup_off, dup_off+4) = fp16(hi * 2^-10), for both CFG rows.  lo_off, dup_off: multiples of 4. */
ew_status ew_euler_cfg_step_split(const void* eps, int ld_eps, float* latents, const float* guidance, float sigma,
                                  float sigma_next, void* next_in, int cpad, int lo_off, int dup_off, int T, int h, int w,
                                  void* stream);

/* Row softmax for the VAE's single-head attention (AutoencoderKLTemporalDecoder mid blocks, head_dim 512: diffusers
 * Attention with upcast_softmax; pipeline_evoworld.py:307-328,358-385 via vae.encode / vae.decode): the [S,S] score matrix
 * of one frame comes out of ew_gemm_f16 split (hi fp16 + lo8 int8, so no fp16 rounding before the exponential);
 * out[r][:] = softmax(decode(hi[r][:], lo[r][:])) in fp32, stored fp16.  lo (element stride ld) may be NULL.  cols % 8 == 0. */
ew_status ew_softmax_rows_f16(const void* hi, const void* lo, void* out, long long rows, int cols, long long ld, void* stream);

/* TemporalDecoder.time_conv_out: Conv3d(C, C, (3,1,1), padding (1,0,0)) over fp32 frames x [B,T,C,HW] (C <= 4; w [C,C,3],
 * bias [C]); zero padding along T.  Replaces the last op of diffusers TemporalDecoder.forward. */
ew_status ew_time_conv3_f32(const float* x, const float* w, const float* bias, float* y, int B, int T, int C, int HW, void* stream);

/* CLIP image-encoder path (pipeline_evoworld.py:255-305 `_encode_image`, :746-850
 * `_resize_with_antialiasing`).  All fp32 planes [planes, H, W]:
 *   ew_blur_axis_f32       1-D correlation along H (axis 0) or W (axis 1) with torch 'reflect' padding, pad_front = (ksize-1)/2
 *                          (`_filter2d` with a [1,k] / [k,1] Gaussian)
 *   ew_bicubic_resize_f32  F.interpolate(mode="bicubic", align_corners=True) (A = -0.75), optional per-channel affine
 *                          out = v*scale[c] + shift[c] (folds (x+1)/2 and the CLIP mean / std normalisation)
 *   ew_vit_patchify_f16    pixel_values [N,3,S,S] -> fp16 [N*(S/P)^2, ldk] im2col of the stride-P patch embedding, K order (c,ky,kx)
 *   ew_attn_small_f16      softmax(q k^T * scale) v for short sequences (S <= 2048) and any head_dim % 8 == 0 (ViT-H: 257 x 80),
 *                          one wave per query, fp32 math; q,k,v token-major rows with stride ld, head h at +h*D. */
ew_status ew_blur_axis_f32(const float* x, const float* kern, int ksize, float* out, long long planes, int H, int W, int axis,
                           void* stream);
ew_status ew_bicubic_resize_f32(const float* x, float* out, int N, int C, int H, int W, int Ho, int Wo, const float* scale,
                                const float* shift, void* stream);
ew_status ew_vit_patchify_f16(const float* x, void* out, int N, int S, int P, int ldk, void* stream);
ew_status ew_attn_small_f16(const void* q, const void* k, const void* v, void* o, int n_seq, int S, int heads, int D, int ld,
                            int ld_o, float scale, void* stream);

/* (ABI 3-9 exported ew_quant_rows_fp8 / ew_gemm_fp8: optional fp8 (OCP e4m3) q / k / v projections for BASELINE.json configs[4].  Measured slower
 * than the fp16 generation-3 GEMMs in rounds 2-4 (2062 vs 1950 ms per forward at the configs[4] size; the projections are HBM- / epilogue-bound,
 * halving operand bytes buys nothing) and 6x outside the parity tolerance: removed from the library in ABI 10 -- configs[4] runs fp16.  The
 * kernels and their test live on under tools/experiments/fp8_qkv/.) */

/* Plücker embedding: out[n, 0:3, y, x] = R_n d(y,x); out[n, 3:6] = t_n x (R_n d)  (fp32).
 * rays [H,W,3] fp32, c2w [N,3,4] fp32 -> out [N,6,H,W] fp32.
 * Replaces utils/plucker_embedding.py:221-255 (ray_c2w_to_plucker). */
ew_status ew_plucker_embed(const float* rays, const float* c2w, float* out, int N, int H, int W, void* stream);

/* Cube -> equirect gather through the integer LUT (face, v, u) int16 [H,W,3]; faces uint8
 * [V,6,res,res,face_channels] (face order right,left,bottom,top,front,back; 3 = packed RGB, 4 = RGBX words) -> pano uint8
 * [V,H,W,3].  Four pixels per thread, dword stores; the LUT is decoded once for all V views.
 * Replaces CubemapRenderer.cube_to_equirectangular_cuda, reproject_vggt_open3d_utils.py:542-614. */
ew_status ew_cube2equi_gather(const uint8_t* faces, int face_channels, const int16_t* lut, uint8_t* pano, int V, int H,
                              int W, int res, void* stream);

/* Depth lift: xyz[s,y,x] = R_s^T (K_s^-1 [u,v,1] z - t_s); depth [S,H,W] f32, extr [S,3,4] world->cam,
 * intr [S,3,3] -> xyz [S,H,W,3] f32.  Replaces vggt unproject_depth_map_to_point_map
 * (unified_loop_consistency.py:352,365-367). */
ew_status ew_depth_unproject(const float* depth, const float* extr, const float* intr, float* xyz, int S, int H, int W,
                             void* stream);

/* Percentile filter of the lifted point cloud (PointCloudProcessor.filter_predictions / _apply_confidence_filter,
 * reproject_vggt_open3d_utils.py:174-222,294-310): np.percentile(conf, q) needs the two order statistics around the
 * virtual index; ew_select_kth_f32 finds x_(k) and x_(k+1) (0-based, ascending) of n floats by an MSD radix select
 * (4 histogram passes + 1 tail pass, no sort) and writes them to out2[0..1] (device).  ws: ew_select_workspace_bytes().
 * ew_filter_compact keeps the points with conf >= thr IN ORDER (boolean-mask semantics): xyz [n,3] f32 -> out_xyz, colours
 * (images * 255 truncated to uint8, :286-292) -> out_rgbx (one R|G<<8|B<<16 word per point); *total = number kept.
 * img: [n,3] floats (img_nchw = 0) or [S,3,hw] planes (img_nchw = 1, n = S*hw).  ws: ew_filter_compact_workspace_bytes(n). */
size_t ew_select_workspace_bytes(void);
ew_status ew_select_kth_f32(const float* x, size_t n, size_t k, void* ws, float* out2, void* stream);
size_t ew_filter_compact_workspace_bytes(size_t n);
ew_status ew_filter_compact(const float* conf, size_t n, float thr, const float* xyz, const float* img, int img_nchw,
                            unsigned hw, float* out_xyz, unsigned* out_rgbx, void* ws, unsigned* total, void* stream);

/* Point splat into cubemap z-buffers: for view v, face f: p_cam = w2c[v][f] * p; u = fx*x/z+cx, ...;
 * nearest pixel, min depth wins (64-bit atomicMin of depth-bits<<32 | point index -> deterministic winner), z > near.
 * Every point is read once (16-byte loads) and tested against all V*6 matrices (wave-uniform, in SGPRs); fragments that a
 * relaxed read of the cell already beats skip the atomic.  zbuf: uint64 [V,6,res,res], 16-byte aligned, UNINITIALISED on entry: since
 * ABI 10 the call itself sets every cell to 0xFF..FF ("no fragment") before the splat (ABI <= 9: the caller pre-filled it).
 * ew_splat_resolve writes the winners' colours (0 background): rgb = packed RGB bytes (rgb_stride 3) or RGBX words
 * (rgb_stride 4, as ew_filter_compact emits); faces uint8 [V,6,res,res,face_channels] (3 | 4).
 * Replaces Open3D OffscreenRenderer point rendering driven by render_face/render_cubemap,
 * reproject_vggt_open3d_utils.py:617-666 (parity unpinned: Filament GL). */
ew_status ew_splat_cubemap(const float* xyz, size_t npts, const float* w2c, unsigned long long* zbuf, int V, int res,
                           float fx, float fy, float cx, float cy, float z_near, void* stream);
ew_status ew_splat_resolve(const unsigned long long* zbuf, const uint8_t* rgb, int rgb_stride, uint8_t* faces,
                           int face_channels, int V, int res, void* stream);

/* Equirect -> perspective bilinear gather (pyequilib Equi2Pers restatement; unified_loop_consistency.py:299-334).
 * equi uint8 [F,He,We,3]; rot [F,3,3] f32 (camera->pano rotation); out uint8 [F,Hp,Wp,3]. */
ew_status ew_equi2pers(const uint8_t* equi, const float* rot, uint8_t* out, int F, int He, int We, int Hp, int Wp,
                       float fov_x_deg, void* stream);

/* Pillow-exact antialiased bilinear resize of 8-bit RGB images (two fixed-point passes, horizontal then vertical, 8-bit
 * intermediate), i.e. what torchvision.transforms.Resize does to the PIL memory panoramas before they enter the pipeline
 * (dataset/CameraTrajDataset.py:586-619 via unified_loop_consistency.py:422).  Coefficient tables kk [n_out, ksize] (int32,
 * 22 fractional bits) and bounds [n_out, 2] = (xmin, count) are built on the host (evoworld_amd.reprojection.resample_coeffs).
 * src [V,Hi,Wi,3] -> tmp [V,Hi,Wo,3] -> dst [V,Ho,Wo,3].  ew_u8_hwc_to_f32_chw: (x/255)*2-1 -> fp32 [V,3,H,W]. */
ew_status ew_resize_aa_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, const int* kk_h, const int* bounds_h, int ksize_h,
                          const int* kk_v, const int* bounds_v, int ksize_v, int V, int Hi, int Wi, int Ho, int Wo, void* stream);
ew_status ew_u8_hwc_to_f32_chw(const uint8_t* src, float* dst, int V, int H, int W, void* stream);
/* The 8-bit frame the reference carries between segments: fp32 [V,3,H,W] in [-1,1] -> u8 [V,H,W,3] =
 * round_half_even(clamp(x/2+0.5, 0, 1)*255), i.e. the pipeline's PIL output (pipeline_evoworld.py:727-732 via diffusers
 * VideoProcessor) that unified_loop_consistency.py:418-419,432-436 feeds to the next segment and to pano->pers. */
ew_status ew_f32_chw_to_u8_hwc(const float* src, uint8_t* dst, int V, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif
