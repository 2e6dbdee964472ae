// Shared structs and device helpers of the convolution translation units (conv.hip = dispatch + C ABI;
// conv_f32.hip, conv_wgrad_f32.hip, conv_sp_im2col.hip, conv_sp_patch.hip, conv_sp_pgroup.hip, conv_ws.hip,
// conv_wgrad_sp.hip = the kernel families and their launchers).  The split exists for build time only: every
// launcher below is a plain function that picks the template instance for a plan the dispatcher made.
#pragma once
#include "common.h"
#include <string.h>
#include <stdint.h>

// ---------------------------------------------------------------------------
struct IgemmArgs {
  const float* x;     // input pixels  [B,Hi,Wi,K]  row stride ldx
  const float* w;     // weights [N][T][K]
  const float* bias;  // [N] or null
  float* y;           // output [B,Hy,Wy,N] row stride ldy
  int ldx, ldy;
  int B, Hi, Wi, K;          // K = input channels of this GEMM
  int Ho, Wo, N;             // iteration grid (GEMM rows = B*Ho*Wo), N = output channels
  int M;                     // B*Ho*Wo
  int sy, sx;                // input coord = o*s + off[t]
  int Hy, Wy, oys, oxs, oy0, ox0;  // output pixel = (oy*oys+oy0, ox*oxs+ox0) in Hy x Wy
  int ntaps;
  unsigned long long offy_pk, offx_pk, wtap_pk;  // 4 bits per tap: off+8, off+8, weight tap
  int T;                     // taps stored per weight row (1 or 9)
  int accumulate;            // y += result
  float rcp_hw, rcp_w;       // 1/(Ho*Wo), 1/Wo: exact index division for < 2^24 pixels (fdiv)
  int direct_out;            // output grid == iteration grid: output pixel index = GEMM row
  const float* xmax;         // fp16x2: device scalar max|x| of the pixel operand when it is a gradient (else null)
  float wscale, wscale_inv;  // fp16x2: fixed power-of-two scale of the weight operand and its inverse (1 otherwise)
  int oy_min, ox_min;        // smallest tap offsets (<= 0 for padded convs): the split-precision body bases its descriptor there
  const unsigned char* wimg; // wave-specialised patch body: pre-split weight image (sp_weight_image_kernel), else null
  int cv_w1, cv_nb;          // wave-specialised body, canvas mode: image pitch on the canvas (W + 1; 0 = plain per-image tiling), images
  unsigned cv_magic;         //   ceil(2^32 / cv_w1): canvas column -> image by multiply-high
  const float* res;          // fused epilogue (inference, BatchNorm folded into w / bias): y = relu?(conv + bias + res[pixel][channel])
  int ldr, relu;             //   res may be null; both are ignored by split-K launches (the planners keep ksplit = 1 when set)
  int epi_early;             // wave-specialised body: issue the epilogue's reads (accumulate / residual) ahead of the tile's last slab
  double* stat_partial;      // wave-specialised body, forward: BatchNorm partial sums of the output, one row [2][N] per block (else null)
  int* stat_rows;            // HOST pointer: the launcher writes the number of rows (blocks) there; not read by any kernel
  int w_persistent;          // w is a parameter the caller keeps images of up to date (hrseg_weight_images_refresh): see conv.hip
  int exp_nosplit;           // MEASUREMENT ONLY (hrseg_tune exp_nosplit_x): the wave-specialised producers stage x without splitting it
  int x_presplit;            // x is stored pre-split (hrseg_conv_shape_t.x_split): only the wave-specialised forward body reads that form
};

// exact n / d for 0 <= n < 2^24 via the float reciprocal (+-1 correction); rcp <= 0 (set by the host for
// tensors of 2^24 pixels or more) selects the plain integer division
__device__ __forceinline__ int fdiv(int n, int d, float rcp) {
  if (rcp <= 0.f) return n / d;
  int q = (int)((float)n * rcp);
  const int r = n - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// Global -> register staging goes through BUFFER loads (resource descriptor in SGPRs + one 32-bit
// byte offset per lane) instead of flat global loads with 64-bit per-lane addresses: measured on
// MI355X every vector-memory instruction issued next to an MFMA stream costs matrix-pipe time
// (tools/ubench/mfma_vmem.hip: 6 global loads per 24 MFMAs 140 -> 103 TFLOP/s, as buffer loads
// 115), and the conv kernels gain 10-13 % (tools/ubench/depth_lab.hip).  The descriptor's range
// check also gives the zero padding for free: a lane outside the image uses offset 0xFFFFFFFF.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define HRSEG_BUF_FLAGS 0x00020000      // raw buffer, 32-bit data format (gfx9 family word 3)
#define HRSEG_BUF_OOB 0xFFFFFFFFu
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0,
                                           (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes), HRSEG_BUF_FLAGS);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff_bytes) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff_bytes, 0));
}
// ... and stores: a lane whose offset is HRSEG_BUF_OOB writes nothing (no exec-mask branch around the store)
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff_bytes, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)voff, soff_bytes, 0);
}

// swizzle of the 16-byte slot inside a 64-byte LDS row so that every 16-lane
// group of a ds_read_b128 fragment read hits 16 distinct slots of the bank row
__device__ __forceinline__ int lds_slot(int row, int slot) { return slot ^ ((-(row >> 2)) & 3); }


// Several independent convolutions (the parallel HRNet branches) in ONE grid: block ranges
// [blk_end[g-1], blk_end[g]) belong to problem g, each with its own tile count and split-K factor.
#define MAXG 8
struct IgemmGroup {
  int n;
  int blk_end[MAXG];
  int tiles[MAXG];   // m-tiles * n-tiles of problem g (its blocks = tiles * ksplit)
  int ksplit[MAXG];
  int kind[MAXG];    // split-precision groups: 1 = halo-patch body (3x3 stride 1 on a wide image), 0 = im2col body
  IgemmArgs a[MAXG];
};
// weight-gradient problem (kernels further down)
struct WgradArgs {
  const float* x;   // [B,Hi,Wi,Cin] ldx
  const float* dy;  // [B,Ho,Wo,Cout] lddy
  float* dw;        // [Cout][T][Cin]
  int ldx, lddy;
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int M;            // B*Ho*Wo
  int ks, stride, T;
  int pix_per_block;  // multiple of the stage size
  float rcp_hw, rcp_w;  // 1/(Ho*Wo), 1/Wo
  const float* dymax;   // fp16x2: device scalar max|dy| (null: unscaled)
};

struct WgradGroup {
  int n;
  int blk_end[MAXG];
  int gx[MAXG];       // pixel-range blocks of problem g (its blocks = gx * tiles)
  WgradArgs a[MAXG];
};
struct IgemmPlan { int wtm, wtn, kc, db, ksplit; };
struct SpPlan { int wtm, wtn, ksplit; };

// ---- launchers (one translation unit per kernel family); return 0, 1 = no instance for the plan
int launch_igemm_f32(const IgemmArgs& a, const IgemmPlan& pl, hipStream_t st);                       // conv_f32.hip
int launch_igemm_group_f32(const IgemmGroup& g, int wtm, int wtn, int kc, hipStream_t st);
int launch_wgrad_f32(const WgradArgs& a, int tn, int tk, int pix, int db, int target, hipStream_t st);   // conv_wgrad_f32.hip
int launch_wgrad_group_f32(const WgradGroup& g, int tn, int tk, int nblocks, hipStream_t st);
int launch_sp_kernel(int ns, const IgemmArgs& a, const SpPlan& pl, hipStream_t st);                  // conv_sp_im2col.hip
int launch_spw_kernel(const IgemmArgs& a, int wtn, int ksplit, unsigned char* img, hipStream_t st);
int launch_sp_group_kernel(int ns, const IgemmGroup& g, int wtm, int wtn, bool full, hipStream_t st);
int launch_patch_sp_kernel(int ns, const IgemmArgs& a, int wtn, int cs, int flip, int blocks, int ntotal, hipStream_t st);  // conv_sp_patch.hip
int launch_sp_pgroup_kernel(int ns, const IgemmGroup& g, int wtm, int wtn, int cs, int flip, hipStream_t st);               // conv_sp_pgroup.hip
int launch_ws_kernel(const IgemmArgs& a, int kind, int flip, int blocks, int ntotal, hipStream_t st, int ns = 4);    // conv_ws.hip
int launch_ws_group_kernel(const IgemmGroup& g, int flip, hipStream_t st, int ns = 4);
int launch_weight_images(const struct WeightImageGroup& g, int nblocks, hipStream_t st);
int launch_weight_image_table(const struct WeightImageTabEntry* tab, int n, int nblocks, hipStream_t st);
int launch_wgrad_sp_kernel(int ns, const WgradArgs& a, int tn, int tk, int gx, int tiles, hipStream_t st);     // conv_wgrad_sp.hip
int launch_wgrad_group_sp(const WgradGroup& g, int tn, int tk, int nblocks, hipStream_t st);
int launch_wgrad_spw_kernel(const WgradArgs& a, int gx, int tiles, hipStream_t st);
int launch_wgrad9_kernels(int ns, int tnk, const struct Wgrad9Group& g, int nblocks, const struct Wgrad9Reduce& r, int rblocks, hipStream_t st);
int check_wgrad_span(const WgradArgs& a);
#define HRSEG_SMALL_CIN_MAX 8
void launch_small_cin_fwd(const float* x, const float* w, const float* bias, float* y, const hrseg_conv_shape_t* s, hipStream_t st);  // conv_small.hip
void launch_small_cin_wgrad(const float* x, const float* dy, float* dw, const hrseg_conv_shape_t* s, int pix_per_block, hipStream_t st);
void launch_small_cin_dgrad(const float* dy, const float* w, float* dx, int accumulate, const hrseg_conv_shape_t* s, hipStream_t st);                                                              // conv.hip
