/* hrseg.h -- C ABI of libhrseg_hip.so: the MI355X (gfx950) kernels behind the
 * hierarchical-segmentation train-step hot path.
 *
 * The reference (Banksylel/Restrictive-Hierarchical-Semantic-Segmentation) has
 * no FFI layer: its hot path is eager PyTorch ops called from
 * Models/models.py, Metrics/losses.py, Metrics/performance_metrics.py and
 * train.py.  Each entry point below replaces the PyTorch/cuDNN op(s) cited
 * next to it; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the
 *     caller (PyTorch's caching allocator); the library allocates nothing.
 *   - process-wide state (all of it listed here): a thread-local error string;
 *     the tuning overrides of hrseg_tune (plain ints, set them before launching
 *     from several threads); the scratch buffer attached with hrseg_set_scratch
 *     (ONE buffer per process, bound to the device that was current when it was
 *     attached -- launches on another device do not use it -- whose region table
 *     is mutex-protected); the weight-image arena attached with
 *     hrseg_set_weight_image_arena and its host-side table of cached weights (one
 *     per process and device, touched only from the launching thread); the launch
 *     counters of hrseg_launch_count.
 *   - activations are NHWC fp32, addressed as pixel*ld + channel ("ld" =
 *     floats per pixel row, >= C, multiple of 4) so a kernel can read or
 *     write a channel slice of a wider tensor (concat without a copy).
 *   - conv weights are OHWI fp32 = torch channels_last storage of the
 *     reference's [Cout,Cin,kh,kw] parameter.
 *   - logits / probabilities / targets at the API boundary are NCHW fp32
 *     contiguous, as the reference returns them.
 *   - all launches go to `stream` (a hipStream_t); no call synchronises.
 *   - return 0 on success, a negative hrseg_status otherwise;
 *     hrseg_last_error_string() describes the last failure on this thread.
 */
#ifndef HRSEG_H
#define HRSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hrseg_stream_t; /* hipStream_t */

enum hrseg_status {
  HRSEG_OK = 0,
  HRSEG_ERR_INVALID_ARG = -1,
  HRSEG_ERR_LAUNCH = -2,
  HRSEG_ERR_UNSUPPORTED = -3,
};

const char* hrseg_last_error_string(void);
int hrseg_abi_version(void);

/* ------------------------------------------------------------------ convolution
 * Replaces nn.Conv2d as used at Models/models.py:113,116 (UNet 3x3+bias),
 * :322-324 conv3x3, :364-370 Bottleneck 1x1/3x3, :483-511 fuse 1x1 / 3x3 s2,
 * :579-582 stem, :614 shared_head 1x1, :656 downsample, :690,:701 transitions.
 * k in {1,3}, stride in {1,2}, pad = (k-1)/2.  Cin, Cout multiples of 16
 * (implicit GEMM on v_mfma_f32_16x16x4_f32) except the stem-style Cin<=4 layer,
 * which hrseg_conv_fwd/wgrad route to direct kernels.                        */
/* Arithmetic of the contraction (operands and results are fp32 in memory in every mode):
 *   F32     fp32 operands on v_mfma_f32_16x16x4_f32 (exact fp32 products)
 *   BF16X3  each operand split exactly into 3 bf16 pieces, the 6 largest of the 9 piece products on
 *           v_mfma_f32_16x16x32_bf16 with fp32 accumulation: fp32-grade (dropped terms < 2^-24 relative)
 *           at 16/6 of the fp32 matrix rate
 *   BF16X2  2 pieces, 3 products (operand error 2^-16)
 *   BF16    operands rounded to bf16, fp32 accumulation (BASELINE configs[4] arithmetic)
 *   AUTO    fp32-grade results from the faster family per problem: FP16X2 for forward / data-gradient problems of
 *           at least 8192 output pixels, for every 3x3 stride-1 problem the wave-specialised kernels take (>= 96
 *           tiles) and for every weight gradient; F32 for the remaining small problems
 *   FP16X2  each operand scaled by a power of two and split into 2 fp16 pieces (22 significand bits), the 3
 *           largest piece products on v_mfma_f32_16x16x32_f16 with fp32 accumulation: operand error 2^-22 at
 *           half the matrix work of BF16X3.  Weights are scaled by 2^8, gradient operands by 2^14 / |max|
 *           (`grad_absmax`, written by hrseg_bn_bwd_group); activations are NOT scaled and NOT clamped:
 *           |x| <= 65504 is exact to 22 bits, up to ~1.3e5 the low piece absorbs the excess, beyond that the
 *           result is Inf / NaN (loud, never a silently saturated value); NaN / Inf inputs propagate.  A caller
 *           that cannot bound its activations checks them with hrseg_absmax and passes F32 (ops.py does so in
 *           deterministic mode).                                                                             */
enum hrseg_conv_precision { HRSEG_CONV_F32 = 0, HRSEG_CONV_BF16X3 = 1, HRSEG_CONV_BF16X2 = 2, HRSEG_CONV_BF16 = 3, HRSEG_CONV_AUTO = 4,
                            HRSEG_CONV_FP16X2 = 5 };
typedef struct {
  int B, Hi, Wi, Cin, ldx; /* input  x[B,Hi,Wi,Cin], row stride ldx   */
  int Ho, Wo, Cout, ldy;   /* output y[B,Ho,Wo,Cout], row stride ldy  */
  int ksize, stride;       /* 1 or 3 ; 1 or 2                          */
  int precision;           /* hrseg_conv_precision                     */
  const float* grad_absmax; /* FP16X2, data / weight gradient: DEVICE array of 64 floats whose maximum is max|dy|
                               of the gradient operand (hrseg_bn_bwd_t.dy_absmax writes it; NULL: unscaled --
                               gradients below 6e-5 then lose precision)                                   */
  const float* residual;    /* forward only, fused epilogue for inference with BatchNorm folded into the weights       */
  int ldr, relu;            /* (hrseg_bn_fold): y = relu?(conv(x, w) + bias + residual[pixel*ldr + channel]); residual
                               may be NULL; zero-initialised fields = plain convolution                               */
  double* stat_partial;     /* forward only, optional: DEVICE buffer of 256 * 2 * Cout doubles.  A kernel that can (the
                               wave-specialised 3x3 kernels) leaves the BatchNorm partial sums of its OUTPUT there -- row r
                               = [sum y over the pixels block r produced][sum y^2], per output channel -- the layout
                               hrseg_bn_fwd_t.partial has with nchunks = *stat_rows: the caller then runs hrseg_bn_fwd_group_
                               phases without the statistics phase (phases 6).  The rows are complete sums in a fixed order
                               per block, but the blocks' LDS atomics meet in any order (last bits of the fp64 sums vary): not
                               offered in deterministic mode                                                              */
  int* stat_rows;           /* HOST pointer, out (set by the call, synchronously): rows written to stat_partial; 0 = the
                               kernel that ran produces no statistics (run the statistics phase).  Required with stat_partial */
  int x_split;              /* forward / weight gradient: x is stored PRE-SPLIT (hrseg_bn_fwd_t.z_split wrote it: per 4 channels the
                               dwords {hi01, hi23, lo01, lo23} of the fp16x2 split).  Only the wave-specialised forward kernels and
                               the nine-tap weight gradient read that form: ask hrseg_conv_x_split_ok first; a call that cannot
                               take those kernels FAILS (HRSEG_ERR_UNSUPPORTED), it never reads the bytes as fp32           */
  int w_persistent;         /* forward / data gradient: the weight operand is a parameter inside one of the ranges registered
                               with hrseg_set_weight_image_arena, and the caller calls hrseg_weight_images_refresh after every
                               change of those parameters: the kernels that read pre-split weight images then take the cached
                               image instead of writing one in front of the launch.  0 (default): never cached            */
} hrseg_conv_shape_t;

/* 1 when the forward call hrseg_conv_fwd(_group)(n, shapes) would run the wave-specialised kernels for EVERY problem and
 * hrseg_conv_wgrad_group_ws would run the nine-tap kernel for all of them, i.e. when the producer of x may store it pre-split
 * (x_split); 0 otherwise.  Pure host logic on the shapes, the attached scratch / tuning state and the precision.           */
int hrseg_conv_x_split_ok(int n, const hrseg_conv_shape_t* shapes);
/* y = conv(x, w) + bias.  w: [Cout][k*k][Cin]; bias may be NULL. */
int hrseg_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                   const hrseg_conv_shape_t* s, hrseg_stream_t stream);
/* dx = conv_transpose(dy, w) (+ dx if accumulate).  wt: [Cin][k*k][Cout]
 * (hrseg_weight_transpose of w).  Shape as the forward conv's. */
int hrseg_conv_dgrad(const float* dy, const float* wt, float* dx, int accumulate,
                     const hrseg_conv_shape_t* s, hrseg_stream_t stream);
/* dw += x (*) dy  (fp32 atomics; dw is [Cout][k*k][Cin] and must hold the
 * running gradient, zeroed by the caller at the start of a step). */
int hrseg_conv_wgrad(const float* x, const float* dy, float* dw,
                     const hrseg_conv_shape_t* s, hrseg_stream_t stream);
/* all conv weights of a flat parameter buffer at once: `table` (device, 8 ints per entry) lists one
 * 32x32 (cout,cin) tile of one tap per entry {slot offset, Cout, taps, Cin, co0, ci0, tap, 0}; the
 * transposed weight of a slot is written at the same offset of flat_t. */
int hrseg_weight_transpose_all(const float* flat, float* flat_t, const int* table, int nentries,
                               hrseg_stream_t stream);
/* Grouped forms: n independent convolutions (the parallel HRNet branches, models.py:524-525; the
 * fuse paths of a module, :483-511) in ONE launch when n <= 8 and they can share a kernel instance
 * (channel counts all multiples of 48 or all of 64), else n separate launches; a stride-2 data
 * gradient is one launch per problem (its four parity classes grouped).  Problems of one data-gradient
 * call must write distinct dx buffers.  Pointer arrays are HOST arrays. */
int hrseg_conv_fwd_group(int n, const float* const* x, const float* const* w,
                         const float* const* bias, float* const* y,
                         const hrseg_conv_shape_t* shapes, hrseg_stream_t stream);
int hrseg_conv_dgrad_group(int n, const float* const* dy, const float* const* wt, float* const* dx,
                           const int* accumulate, const hrseg_conv_shape_t* shapes,
                           hrseg_stream_t stream);
int hrseg_conv_wgrad_group(int n, const float* const* x, const float* const* dy, float* const* dw,
                           const hrseg_conv_shape_t* shapes, hrseg_stream_t stream);
/* Weight gradients with a caller-provided workspace: full 3x3 stride-1 problems in a split-precision mode
 * (channels multiples of 48, or of 64) run the nine-tap kernel -- per-block partial sums written with plain
 * stores into the workspace, then added to dw in a fixed order (no atomics: bit-reproducible).
 * hrseg_conv_wgrad_workspace_bytes returns the size that path needs for the n problems (0: it does not
 * apply); with a NULL / too small workspace, or other shapes, the call is hrseg_conv_wgrad_group. */
size_t hrseg_conv_wgrad_workspace_bytes(int n, const hrseg_conv_shape_t* shapes);
int hrseg_conv_wgrad_group_ws(int n, const float* const* x, const float* const* dy, float* const* dw,
                              const hrseg_conv_shape_t* shapes, void* workspace, size_t workspace_bytes,
                              hrseg_stream_t stream);
/* launches issued so far by kernel family ("ws", "ws_group", "patch_sp", "sp_im2col", "sp_pgroup", "sp_group", "f32",
 * "f32_group", "wgrad_sp", "wgrad_sp_group", "wgrad_f32", "wgrad_f32_group", "wgrad9", "small_cin", "sp_wide"; NULL = all); reset != 0
 * zeroes what it returns.  "ws_canvas" counts PROBLEMS (not launches, not part of the NULL total) that a "ws" / "ws_group"
 * launch tiled as one canvas of side-by-side images, "wgrad_sp_t5" the "wgrad_sp" launches on 80 x 80 tiles ("wgrad_sp_wide": the wide-tile weight-gradient kernel, a family of its own).  The parity tests use it to prove which kernels a case ran. */
long hrseg_launch_count(const char* family, int reset);
/* tile-plan overrides and A/B switches for the sweep tools under tools/ (value 0 = automatic plan).  Keys: igemm_wtm,
 * igemm_kc, igemm_db, igemm_ksplit, group_wtm, wgrad_pix, wgrad_db, wgrad_blocks, wgrad_group_mult,
 * wgrad_group_min, wgrad_group_max (fp32 kernels); sp_wtm, sp_wtn, sp_ksplit, sp_patch, sp_persist (split-precision
 * kernels); sp_ws (0: never use the wave-specialised 3x3 kernels), sp_ws_n48 (0: 48-channel tilings stay on the
 * block-synchronous kernels), sp_ws_waste (accepted tile padding, percent), sp_ws_bf16 (0: the BF16 arithmetic stays off the wave-specialised kernels), small_cin3 (0: the 3-channel first layer on the generic
 * Cin <= 8 kernels), sp_ws_canvas (0: tile every image on its own, never the batch as one
 * canvas), sp_img (0: block-synchronous kernels
 * split their weights on the fly); wgrad9 (0: never the nine-tap weight gradient), wgrad9_blocks (its target block count per problem; wgrad9_blocks1 .. wgrad9_blocks4: the same for launches of 1 .. 4 problems); wgrad_group_sp (0: grouped tap-per-block
 * weight gradients stay on the fp32 kernel), wgrad_sp_t5 (0: no 80 x 80 tiles for the wide layers), wgrad_sp_wide (0: never the wide-tile weight-gradient body); deterministic (1: single-adder
 * reductions everywhere); routing thresholds sp_ws_min_tiles (96), auto_min_pixels (8192), sp_patch_min_tiles (192): the
 * parity tests lower them so that small cases run the kernels of the headline sizes -- csrc/conv.hip, hrseg_tune.  Unknown key: HRSEG_ERR_INVALID_ARG. */
int hrseg_tune(const char* key, int value);
/* Persistent weight images.  The wave-specialised kernels read their weights from a pre-split image (4 bytes per weight); by
 * default a small kernel writes it in front of every launch.  With an arena attached -- DEVICE memory the caller owns: `arena`
 * (256-byte aligned, at least 1 MiB; 8 bytes per parameter covers forward and data-gradient images of a whole model) and
 * `table` (256-byte aligned, 40 bytes per cached weight) -- the images of weights flagged hrseg_conv_shape_t.w_persistent that lie
 * in [lo0, hi0) or [lo1, hi1) (the parameter buffer and its transposed copy) are kept: written on first use, and ALL rewritten
 * by ONE launch of hrseg_weight_images_refresh, which the caller issues whenever those parameters changed (once per train step)
 * before the convolutions.  Re-attaching (or (NULL, 0, ...)) forgets every cached image.  One arena per process, bound to the
 * device that was current when it was attached.  Reference ops replaced: none (an implementation detail of nn.Conv2d here). */
int hrseg_set_weight_image_arena(void* arena, size_t bytes, void* table, size_t table_bytes, const float* lo0, const float* hi0,
                                 const float* lo1, const float* hi1);
int hrseg_weight_images_refresh(hrseg_stream_t stream);
/* Scratch memory for the convolution launches (device, 256-byte aligned, at least 1 MiB; 256 MiB covers every layer
 * of the reference's models): the wave-specialised fp16x2 kernels of the wide 3x3 stride-1 layers read their weights
 * from a pre-split image (4 bytes per weight) that a small kernel writes there right before each launch, on the same
 * stream (up to eight streams get an eighth of the buffer each, as a ring; the images of one grouped launch are reserved
 * together, so the ring never wraps inside a group; a group whose images do not fit a region, a ninth stream, or a launch
 * on another device than the one current at attach time takes the block-synchronous kernels).  The buffer stays the caller's and must outlive the launches; (NULL, 0) detaches it.
 * Without it those layers run the block-synchronous kernels: same results, slower. */
int hrseg_set_scratch(void* ptr, size_t bytes);
/* wt[ci][t][co] = w[co][t][ci] */
int hrseg_weight_transpose(const float* w, float* wt, int Cout, int taps, int Cin,
                           hrseg_stream_t stream);

/* ------------------------------------------------------------------ batch norm
 * Replaces nn.BatchNorm2d / SyncBatchNorm-without-process-group in training
 * mode (Models/models.py:114,117; bn_helper.py:4-11; BN_MOMENTUM :318) fused
 * with the ReLU / residual add that follows it (:115,118,345,353-354,542).   */
/* per-channel partial sums of y and y*y over pixel chunks -> partial[nchunks][2][C] (double) */
int hrseg_bn_stats(const float* y, int ldy, long npix, int C, double* partial, int nchunks,
                   hrseg_stream_t stream);
/* reduce partials; write mean,rstd,scale,shift ([4][C] fp32 in `coef`); update
 * running stats (momentum, unbiased var) and ++num_batches_tracked if given. */
int hrseg_bn_finalize(const double* partial, int nchunks, long npix, int C, const float* gamma,
                      const float* beta, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, float eps, float* coef,
                      hrseg_stream_t stream);
/* inference: BatchNorm (running statistics) folded into the convolution in front of it -- w_out[co][:] = w[co][:] * s,
 * b_out[co] = (bias[co] - running_mean[co]) * s + beta[co] with s = gamma[co] / sqrt(running_var[co] + eps); `row` =
 * k*k*Cin floats per output channel (OHWI), bias may be NULL.  With hrseg_conv_shape_t.residual / relu the whole
 * conv + BN (+ residual) (+ ReLU) of models.py:113-118, 332-354 is then ONE launch. */
int hrseg_bn_fold(const float* w, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                  const float* running_var, float eps, int Cout, int row, float* w_out, float* b_out,
                  hrseg_stream_t stream);
/* eval mode: coef from running stats */
int hrseg_bn_eval_coef(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* coef,
                       hrseg_stream_t stream);
/* z = relu?( y*scale + shift (+ residual) ) */
int hrseg_bn_apply(const float* y, int ldy, const float* coef, const float* residual, int ldr,
                   int relu, float* z, int ldz, long npix, int C, hrseg_stream_t stream);
/* backward, phase 1: g = dz * (z>0 if relu); partial sums of g and g*xhat */
int hrseg_bn_bwd_reduce(const float* dz, int lddz, const float* z, int ldz, int relu,
                        const float* y, int ldy, const float* coef, long npix, int C,
                        double* partial, int nchunks, hrseg_stream_t stream);
/* backward, phase 2: dgamma += sum g*xhat, dbeta += sum g (if not NULL);
 * dy = gamma*rstd*(g - mean_g - xhat*mean_gx); optionally dres (+)= g.
 * `partial` must hold (nchunks+1)*2*C doubles: the last [2][C] receives the totals. */
int hrseg_bn_bwd_apply(const double* partial, int nchunks, const float* dz, int lddz,
                       const float* z, int ldz, int relu, const float* y, int ldy,
                       const float* coef, const float* gamma, float* dgamma, float* dbeta,
                       float* dy, int lddy, float* dres, int lddres, int dres_accumulate,
                       long npix, int C, int eval_mode, hrseg_stream_t stream);

/* Grouped forms: n (1..8) independent BatchNorm problems in three launches (statistics, finalize,
 * apply; eval: coefficients, apply) resp. (reduce, finalize, apply) for the backward. */
typedef struct {
  const float* y; int ldy; long npix; int C;       /* conv output [npix][C]                      */
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;   /* updated in training  */
  float momentum, eps;
  const float* residual; int ldr; int relu;          /* z = relu?(bn(y) + residual)                */
  float* z; int ldz;
  float* coef;                                       /* out: [4][C] mean, rstd, scale, shift       */
  double* partial; int nchunks;                      /* scratch nchunks*2*C doubles (training)     */
  int stat_div;                                      /* training: the tensor holds stat_div identical
                                                        copies of one pass's images (batched level
                                                        passes); the unbiased-variance factor uses
                                                        npix/stat_div (0 or 1 = plain)               */
  int stat_updates;                                  /* training: times the batch statistics enter
                                                        the running averages / num_batches_tracked
                                                        (0 or 1 = once; L = de-duplicated level
                                                        passes, see Models/models.py)               */
  unsigned char* relu_mask;                          /* optional out, [npix][C/4] bytes: bit j of byte (pixel, q) =
                                                        channel 4q+j passed the ReLU; the backward of a layer WITH a
                                                        residual reads it (hrseg_bn_bwd_t.relu_mask) instead of z    */
  int z_split;                                       /* != 0: z is written PRE-SPLIT for fp16x2 convolutions (per 4
                                                        channels {hi01, hi23, lo01, lo23}, same 16 bytes): only for a
                                                        tensor whose every reader takes hrseg_conv_shape_t.x_split */
  int residual_split;                                /* != 0: `residual` is stored pre-split (z_split of the launch that
                                                        wrote it); the apply phase adds hi + lo, i.e. the fp32 value
                                                        rounded to the 22 bits the convolutions multiply            */
  int stat_ranks;                                    /* cross-rank statistics (opt-in synchronised BN): the
                                                        partial sums were all-reduced over this many ranks of
                                                        equal shards between the statistics and the finalize
                                                        phase, so the statistics cover stat_ranks*npix pixels
                                                        (0 or 1 = this rank only, the reference's behaviour)  */
} hrseg_bn_fwd_t;
int hrseg_bn_fwd_group(int n, const hrseg_bn_fwd_t* problems, int training, hrseg_stream_t stream);
/* the same in phases (bit 0 statistics, bit 1 finalize / eval coefficients, bit 2 apply; 7 = all): a caller that
 * synchronises BatchNorm statistics across ranks runs phase 1, all-reduces `partial`, then runs phases 2|4 with
 * stat_ranks set (the reference's SyncBatchNorm would do this WITH a process group; it never has one) */
int hrseg_bn_fwd_group_phases(int n, const hrseg_bn_fwd_t* problems, int training, int phases, hrseg_stream_t stream);
typedef struct {
  const float* dz; int lddz; const float* z; int ldz; int relu;   /* relu with z == NULL: the forward had
                                                        no residual, the mask is recomputed from y    */
  const float* y; int ldy; const float* coef;
  float* dgamma; float* dbeta;                       /* += (may be NULL)                            */
  float* dy; int lddy;                               /* out (may alias dz)                          */
  float* dres; int lddres; int dres_accumulate;      /* residual gradient (=|+=) g, or NULL         */
  long npix; int C;
  double* partial; int nchunks;                      /* scratch (nchunks+max(nseg,1))*2*C doubles   */
  float* dy_absmax;                                  /* optional DEVICE array of 64 floats (reset by the call):
                                                        slot (block % 64) receives the max|dy| of its blocks;
                                                        feeds hrseg_conv_shape_t.grad_absmax              */
  int nseg;                                          /* > 1: npix is nseg equal segments (the batched
                                                        level passes), each normalised on its own: the
                                                        batch means of the backward are per segment;
                                                        must divide nchunks and npix (0 or 1 = plain)  */
  const unsigned char* relu_mask;                    /* optional: the forward's ReLU mask bytes; given, neither z is
                                                        read nor the mask recomputed (4 B per element less per pass
                                                        on layers with a residual)                                  */
  int sum_ranks;                                     /* cross-rank statistics: the partial sums were all-reduced
                                                        over this many ranks between the reduce and the finalize
                                                        phase; the finalize divides them by it, so that the batch
                                                        means are global and dgamma / dbeta receive this rank's
                                                        share (0 or 1 = this rank only)                      */
} hrseg_bn_bwd_t;
int hrseg_bn_bwd_group(int n, const hrseg_bn_bwd_t* problems, int eval_mode, hrseg_stream_t stream);
/* phases: bit 0 reduce, bit 1 finalize, bit 2 apply (7 = all), see hrseg_bn_fwd_group_phases */
int hrseg_bn_bwd_group_phases(int n, const hrseg_bn_bwd_t* problems, int eval_mode, int phases, hrseg_stream_t stream);

/* ------------------------------------------------------------------ pooling / resampling / glue
 * nn.MaxPool2d(2) (models.py:140); bilinear align_corners=True resize
 * (nn.Upsample :156, F.interpolate :536-539,:746,:757,:766,:776) incl. the
 * zero pad of `up` (:166-170) and the channel concat (:172,:747).            */
int hrseg_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, int B, int Hi, int Wi, int C,
                       hrseg_stream_t stream);
int hrseg_maxpool2_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx,
                       int accumulate, int B, int Hi, int Wi, int C, hrseg_stream_t stream);
/* out[b, py+oy, px+ox, :] (op)= bilinear(in)[oy,ox]; region outside the
 * placed image is zero-filled when !accumulate. out image is Hout x Wout,
 * the resized image Hr x Wr placed at (py,px). relu applied after the add. */
int hrseg_bilinear_fwd(const float* in, int ldin, int B, int Hi, int Wi, int C, float* out,
                       int ldout, int Hout, int Wout, int Hr, int Wr, int py, int px,
                       int align_corners, int accumulate, int relu, hrseg_stream_t stream);
int hrseg_bilinear_bwd(const float* dout, int lddout, int B, int Hi, int Wi, int C, float* din,
                       int lddin, int Hout, int Wout, int Hr, int Wr, int py, int px,
                       int align_corners, int accumulate, hrseg_stream_t stream);
/* max|x| of an NHWC tensor: slot (block % 64) of the 64-float DEVICE array out64 (zeroed by the caller) is raised to
 * the maximum of its blocks; NaN counts as +Inf.  Range check in front of FP16X2 convolutions (see above). */
int hrseg_absmax(const float* x, int ldx, long npix, int C, float* out64, hrseg_stream_t stream);
/* HRNet fuse sum (models.py:527-542) in one pass: out = relu?( sum of n_same (1..4) same-resolution NHWC terms + sum of
 * n_low (0..3) terms bilinearly resized from Hi[j] x Wi[j] to H x W ); HOST arrays of pointers / strides / sizes */
int hrseg_fuse_sum(int n_same, const float* const* same, const int* ld_same, int n_low, const float* const* low,
                   const int* ld_low, const int* Hi, const int* Wi, float* out, int ldo, int B, int H, int W, int C,
                   int align_corners, int relu, hrseg_stream_t stream);
/* out = relu?(a + b) ; strided channel copy ; masked relu backward */
int hrseg_add(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int relu,
              long npix, int C, hrseg_stream_t stream);
int hrseg_copy(const float* in, int ldin, float* out, int ldout, int accumulate, long npix, int C,
               hrseg_stream_t stream);
int hrseg_relu_bwd(const float* dz, int lddz, const float* z, int ldz, float* dx, int lddx,
                   long npix, int C, hrseg_stream_t stream);
int hrseg_nchw_to_nhwc(const float* in, float* out, int ldout, int B, int C, int H, int W,
                       hrseg_stream_t stream);
int hrseg_nhwc_to_nchw(const float* in, int ldin, float* out, int B, int C, int H, int W,
                       hrseg_stream_t stream);

/* ------------------------------------------------------------------ FiLM + heads + composition
 * FiLM (models.py:58-77), outconv / classifier 1x1 heads (:180,:626-645),
 * sigmoid / grouped-softmax probability composition (:266-304,:763-798).     */
/* cond[b,c] = mean_{h,w} P[b,c,h,w]  (NCHW in); scratch: BC*64 doubles */
int hrseg_gap_nchw(const float* p, float* cond, double* scratch, int BC, long hw,
                   hrseg_stream_t stream);
/* gb[b, 0:2F] = cond[b,:] @ Wl^T + bl   (Wl: [2F][Cc]) */
int hrseg_film_linear_fwd(const float* cond, const float* wl, const float* bl, float* gb, int B,
                          int Cc, int F2, hrseg_stream_t stream);
/* dcond = dcond_scale * dgb @ Wl (written); dWl += dgb^T cond; dbl += sum_b dgb */
int hrseg_film_linear_bwd(const float* cond, const float* wl, const float* dgb, float* dcond,
                          float* dwl, float* dbl, int B, int Cc, int F2, float dcond_scale,
                          hrseg_stream_t stream);
/* head: z[b,pix,c] = sum_k W[c][k]*(f[b,pix,k]*gamma[b,k]+beta[b,k]) + bias[c];
 * gb = [B][2F] (gamma | beta) or NULL for no FiLM; z is NHWC with row stride
 * ldz; Cout <= 8. */
int hrseg_head_fwd(const float* f, int ldf, const float* gb, const float* w, const float* bias,
                   float* z, int ldz, int B, long hw, int F, int Cout, hrseg_stream_t stream);
/* backward of the head: df (=|+=) gamma*(W^T dz) (df may be NULL); dW, dbias +=
 * (atomics); dgb[b] += (sum_pix f*(W^T dz) | sum_pix W^T dz) when gb != NULL */
int hrseg_head_bwd(const float* f, int ldf, const float* gb, const float* w, const float* dz,
                   int lddz, float* df, int lddf, int df_accumulate, float* dw, float* dbias,
                   float* dgb, int B, long hw, int F, int Cout, hrseg_stream_t stream);
/* logits resize: NHWC low-res [B,Hi,Wi,C<=16] -> NCHW [B,C,Ho,Wo] bilinear
 * (F.interpolate at models.py:757,766,776) and its transpose */
int hrseg_logits_up_fwd(const float* in, int ldin, int B, int Hi, int Wi, int C, float* out, int Ho,
                        int Wo, int align_corners, hrseg_stream_t stream);
int hrseg_logits_up_bwd(const float* dout, int B, int Hi, int Wi, int C, float* din, int lddin,
                        int Ho, int Wo, int align_corners, hrseg_stream_t stream);
/* P0 = sigmoid(z0) ; NCHW */
int hrseg_sigmoid_fwd(const float* z, float* p, long n, hrseg_stream_t stream);
/* level L>0: groups given as parent channel index + child count per group
 * (group_parent / group_size are HOST arrays, copied into the launch) */
int hrseg_compose_fwd(const float* z, const float* pprev, float* p, int B, int C, int Cprev,
                      long hw, int ngroups, const int* group_parent, const int* group_size,
                      hrseg_stream_t stream);
/* backward through composition: inputs dP (strided: element (b,c,i) at
 * b*sb + c*sc + i*si, so a broadcast GAP gradient needs no materialisation),
 * outputs dz (NCHW, accumulate flag) and dPprev (NCHW, accumulate flag). */
int hrseg_compose_bwd(const float* dp, long sb, long sc, long si, const float* z,
                      const float* pprev, float* dz, int dz_accumulate, float* dpprev,
                      int dpprev_accumulate, int B, int C, int Cprev, long hw, int ngroups,
                      const int* group_parent, const int* group_size, hrseg_stream_t stream);
int hrseg_sigmoid_bwd(const float* dp, long sb, long sc, long si, const float* z, float* dz,
                      int accumulate, int B, int C, long hw, hrseg_stream_t stream);

/* ------------------------------------------------------------------ loss + metrics
 * CrossEntropyLoss / SoftDiceLoss (Metrics/losses.py:16-134), consistency
 * (:150-177), prediction prep + confusion counts (train.py:206-231,
 * Metrics/performance_metrics.py:27-141).                                    */
/* partial[b][c][5] += {n, sum t*logp, sum p*t, sum p, sum t} over t!=-1 (double) */
int hrseg_loss_partials(const float* z, const float* t, double* partial, int B, int C, long hw,
                        hrseg_stream_t stream);
/* out[0]=ce, out[1]=dice, out[2]=dice_valid_count; coef[b][c][4] for the backward */
int hrseg_loss_finalize(const double* partial, const float* w, int B, int C, float* out,
                        float* coef, hrseg_stream_t stream);
/* dz (=|+=) g_ce*dCE/dz + g_dice*dDice/dz ; g = device pointer to {g_ce,g_dice} */
int hrseg_loss_bwd(const float* z, const float* t, const float* coef, const float* g, float* dz,
                   int accumulate, int B, int C, long hw, hrseg_stream_t stream);
/* sum |sum_{c in group} P[b,c] - Pprev[b,parent]| per group -> out[ngroups] (double, +=) */
int hrseg_consistency(const float* p, const float* pprev, double* out, int B, int C, int Cprev,
                      long hw, int ngroups, const int* group_parent, const int* group_size,
                      hrseg_stream_t stream);
/* gradient of  scale * sum_groups sum_{b,pix} |...|  times the device scalar g[0]:
 * dp [B,C,hw] and dpprev [B,Cprev,hw] are written */
int hrseg_consistency_bwd(const float* p, const float* pprev, const float* g, float scale, float* dp,
                          float* dpprev, int B, int C, int Cprev, long hw, int ngroups,
                          const int* group_parent, const int* group_size, hrseg_stream_t stream);
/* Grouped conditional KL, the opt-in stabiliser the reference keeps commented out (Metrics/losses.py:180-210): per parent
 * group of the level Q = softmax_c(z_c + log(Pprev[parent] + 1e-6)).clamp_min(1e-8) over the group's children;
 * out[g] (double, +=) = sum over b, pixels and the group's children of Q * (log Q + log size_g), i.e. KL(Q || Uniform)
 * before the mean.  hrseg_group_kl_bwd writes dz [B,C,hw] = g[0] * scale * d/dz sum_g out[g] / size_g (Pprev gets no
 * gradient: the log-bias is constant inside a group).  Default off in the train loop. */
int hrseg_group_kl(const float* z, const float* pprev, double* out, int B, int C, int Cprev, long hw, int ngroups,
                   const int* group_parent, const int* group_size, hrseg_stream_t stream);
int hrseg_group_kl_bwd(const float* z, const float* pprev, const float* g, float scale, float* dz, int B, int C,
                       int Cprev, long hw, int ngroups, const int* group_parent, const int* group_size,
                       hrseg_stream_t stream);
/* per-class metric vectors of nlevels (1..8) levels from their confusion counts in ONE launch: cm[L] (DEVICE, int64
 * [K_L][K_L], (target, predicted) as hrseg_predict_metrics counts them), child[L] != 0: label 0 is the synthetic background
 * of a child level (its row is dropped, its class not reported).  out (DEVICE) receives [5][sum_L (K_L - child_L)] floats:
 * accuracy (= recall per class, as the reference defines it), iou, dice, precision, recall, levels side by side; a zero
 * denominator gives 0.  Replaces the ~80 small tensor ops of the reference's metric classes per batch (train.py:47-51,
 * Metrics/performance_metrics.py).  cm, K, child are HOST arrays. */
int hrseg_metric_vectors(int nlevels, const long long* const* cm, const int* K, const int* child, float* out,
                         hrseg_stream_t stream);
/* argmax one-hot of z masked by t!=-1, plus confusion matrix counts
 * cm[(C+child)*(C+child)] (int64, +=) of (target label, predicted label) with
 * the synthetic background class 0 for child levels. mask_pred=1 is the train
 * loop (predictions/targets zeroed where t==-1), 0 the test loop (z holds
 * probabilities, raw targets). onehot may be NULL. */
int hrseg_predict_metrics(const float* z, const float* t, float* onehot, long long* cm, int B,
                          int C, long hw, int child, int mask_pred, hrseg_stream_t stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.AdamW (train.py:513-516) over one flat fp32 parameter buffer. */
int hrseg_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                float beta2, float eps, float weight_decay, float bc1, float bc2, float gscale,
                hrseg_stream_t stream);
/* same update with every scalar in DEVICE memory (hipGraph-replayable): hyper = {lr, beta1, beta2,
 * eps, weight_decay, grad_scale}; state = {step, 1-beta1^step, 1/sqrt(1-beta2^step)} is advanced by
 * one step per call. */
int hrseg_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper,
                    float* state, hrseg_stream_t stream);
int hrseg_fill(float* p, float v, long n, hrseg_stream_t stream);

/* ------------------------------------------------------------------ gradient exchange (data parallelism)
 * Thin wrappers over RCCL for the one collective of the path: the in-place sum of a flat fp32 gradient
 * bucket over the ranks (replaces nn.DataParallel's reduce-add, train.py:509-510).  librccl.so is opened
 * on first use.  One communicator per process (= per GPU); the 128-byte id comes from ONE rank and is
 * shipped to the others by the caller (file, TCP store, ...).  All calls are stream-ordered, none
 * synchronises the host. */
int hrseg_comm_unique_id(void* id128);
int hrseg_comm_init(void** comm, int rank, int world, const void* id128);   /* collective over the ranks */
int hrseg_comm_allreduce_async(void* comm, float* buf, long count, hrseg_stream_t stream);
int hrseg_comm_wait(hrseg_stream_t comm_stream, hrseg_stream_t consumer);   /* consumer waits for comm_stream */
int hrseg_comm_destroy(void* comm);

/* ------------------------------------------------------------------ target encoding (input side of the path)
 * SegDataset.separate_masks / traverse_tree / process_ignore_values (Data/dataset.py:41-124,
 * 227-265): label image [B,hw] of uint8 pixel values -> out [B,C,hw] fp32 (NCHW planes).
 * on_lut: DEVICE table [256] of uint64, bit c set when channel c's node contains the leaf class
 * with that pixel value; parent: HOST array [C], channel index of the node's direct parent or -1
 * (roots, and every channel in flat mode).  Channel value: 1 on the node, else 0 for roots / inside
 * the parent's area, else -1.  C <= 64. */
int hrseg_encode_targets(const unsigned char* label, const unsigned long long* on_lut, const int* parent,
                         float* out, int B, int C, long hw, hrseg_stream_t stream);

/* ------------------------------------------------------------------ level synthesis for flat models (evaluation side)
 * predictEval.py:85-129 get_parent_masks (parent = union of its descendant leaves, "any > 0") and :134-185
 * combine_levels (per-level tensors stitched from leaf and parent channels).  out[b][o] is the COPY of the single
 * input channel selected by masks[o] (is_union[o] == 0) or 1.0 where any selected input channel is > 0, else 0.0.
 * Input channel i < C0 is plane i of x0, else plane i - C0 of x1 (x1 may be NULL with C1 == 0).  NCHW planes;
 * masks / is_union are HOST arrays of Cout entries; C0 + C1 <= 64, Cout <= 64. */
int hrseg_combine_levels(const float* x0, int C0, const float* x1, int C1, const unsigned long long* masks,
                         const int* is_union, float* out, int B, int Cout, long hw, hrseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HRSEG_H */
