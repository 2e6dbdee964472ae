// Direct VALU convolution kernels for the first layer (Cin <= 8: the 3-channel image, or image + the previous level's
// logits with logit-concatenated re-encoding): forward, weight gradient, data gradient.  CMAX = 4 or 8 is the compile-time
// channel capacity (the 3-channel layer keeps its 4-wide vectors).
#include "conv_common.h"

int g_small_cin3 = 1;       // hrseg_tune "small_cin3": 0 = the 3-channel 3x3 first layer stays on the generic Cin <= 8 kernels

// y[b,oy,ox,co] = bias[co] + sum_{t,ci} x[b, oy*s+kh-1, ox*s+kw-1, ci] * w[co][t][ci]
// one thread = one output pixel x 16 channels; weights of the block's 64 channels in LDS.
template <int CMAX>
__global__ __launch_bounds__(256) void conv_small_cin_fwd_kernel(const float* __restrict__ x, int ldx,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ y, int ldy, int B, int Hi,
                                                                 int Wi, int Cin, int Ho, int Wo, int Cout,
                                                                 int ks, int stride, const float* __restrict__ res, int ldr,
                                                                 int relu) {
  __shared__ float wl[64 * 9 * CMAX];
  const int T = ks * ks, pad = (ks - 1) / 2;
  const int cb = blockIdx.y * 64;  // channel block
  for (int i = threadIdx.x; i < 64 * T * Cin; i += 256) {
    const int co = i / (T * Cin);
    wl[i] = (cb + co < Cout) ? w[(size_t)(cb + co) * T * Cin + (i - co * T * Cin)] : 0.f;
  }
  __syncthreads();
  const long M = (long)B * Ho * Wo;
  const long m = (long)blockIdx.x * 64 + (threadIdx.x & 63);
  const int cg = threadIdx.x >> 6;  // 16-channel group
  if (m >= M) return;
  const int b = (int)(m / ((long)Ho * Wo));
  const int rem = (int)(m - (long)b * Ho * Wo);
  const int oy = rem / Wo, ox = rem - oy * Wo;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int t = 0; t < T; ++t) {
    const int iy = oy * stride + t / ks - pad, ix = ox * stride + t % ks - pad;
    if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
    const float* xp = x + ((size_t)(b * Hi + iy) * Wi + ix) * ldx;
    for (int ci = 0; ci < Cin; ++ci) {
      const float xv = xp[ci];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(xv, wl[((cg * 16 + j) * T + t) * Cin + ci], acc[j]);
    }
  }
  float* yp = y + (size_t)m * ldy + cb + cg * 16;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int co = cb + cg * 16 + j;
    if (co < Cout) {
      float v = acc[j] + (bias ? bias[co] : 0.f);
      if (res) v += res[(size_t)m * ldr + co];
      yp[j] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

// The actual first layer (3 input channels, 3x3): thread = 4 output channels of one pixel, its 4 x 27 weights in REGISTERS for
// the whole pixel loop (the generic kernel above reads one weight from LDS per FMA and stores 16 scattered floats per thread:
// 264 us for the 197 MB the HRNet stem writes, 6x the time of those bytes).  A wave covers 4 pixels x 64 channels: 256
// contiguous bytes stored per pixel.  Same accumulation order per output element as the generic kernel (tap-major, channel-minor).
__global__ __launch_bounds__(256) void conv_cin3_k3_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                               int B, int Hi, int Wi, int Ho, int Wo, int stride,
                                                               const float* __restrict__ res, int ldr, int relu) {
  const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int co = blockIdx.y * 64 + 4 * cq;
  float wr[27][4];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[k][j] = w[(size_t)(co + j) * 27 + k];          // [co][t][ci], Cin = 3
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + co);
  const int M = B * Ho * Wo, hw = Ho * Wo;
  for (int m = blockIdx.x * 16 + pl; m < M; m += gridDim.x * 16) {
    const int b = m / hw, rem = m - b * hw;
    const int oy = rem / Wo, ox = rem - oy * Wo;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy * stride + kh - 1;
      const bool rok = (unsigned)iy < (unsigned)Hi;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = ox * stride + kw - 1;
        const bool ok = rok & ((unsigned)ix < (unsigned)Wi);
        const float* xp = x + ((size_t)(b * Hi + (ok ? iy : 0)) * Wi + (ok ? ix : 0)) * ldx;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float xv = ok ? xp[ci] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, wr[(kh * 3 + kw) * 3 + ci][j], acc[j]);
        }
      }
    }
    f32x4 v = acc + bv;
    if (res) v += *reinterpret_cast<const f32x4*>(res + (size_t)m * ldr + co);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + (size_t)m * ldy + co) = v;
  }
}

// weight gradient of the same layer: block = 64 output channels x a pixel range, 64-pixel stages through LDS as in the
// generic kernel; thread = (output channel, quarter of the stage's pixels) with ALL 27 (tap, channel) sums in registers: per
// pixel one dy read and nine 16-byte broadcast reads of the staged taps for 27 FMAs (generic kernel: 13 LDS reads per 12 FMAs).
// The four quarters meet in LDS in a fixed order; one atomic add per weight and block.
__global__ __launch_bounds__(256) void conv_cin3_k3_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                                 int lddy, float* __restrict__ dw, int B, int Hi, int Wi, int Ho,
                                                                 int Wo, int stride, int pix_per_block) {
  constexpr int P = 64;
  __shared__ __attribute__((aligned(16))) float dys[P][64];
  __shared__ __attribute__((aligned(16))) float xs[P][9][4];
  __shared__ float red[3][64][28];                      // quarters 1..3 of the block's sums ([28]: bank spread)
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * 64, col = tid & 63, pg = tid >> 6;
  const int M = B * Ho * Wo, hw = Ho * Wo;
  const int lo = blockIdx.x * pix_per_block;
  const int hi = min(lo + pix_per_block, M);
  float acc[9][3];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[t][c] = 0.f;
  for (int s0 = lo; s0 < hi; s0 += P) {
#pragma unroll
    for (int r = 0; r < (P * 16) / 256; ++r) {            // dy tile: thread -> (pixel j>>4, 4 channels (j&15)*4)
      const int j = tid + 256 * r;
      const int pp = j >> 4, c4 = (j & 15) * 4;
      const int m = s0 + pp;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < hi) v = *reinterpret_cast<const f32x4*>(dy + (size_t)m * lddy + co0 + c4);
      *reinterpret_cast<f32x4*>(&dys[pp][c4]) = v;
    }
    for (int j = tid; j < P * 9; j += 256) {              // input taps: item -> (pixel, tap)
      const int pp = j / 9, t = j - pp * 9;
      const int m = s0 + pp;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < hi) {
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const int iy = oy * stride + t / 3 - 1, ix = ox * stride + t % 3 - 1;
        if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
          const float* xp = x + ((size_t)(b * Hi + iy) * Wi + ix) * ldx;
          v = f32x4{xp[0], xp[1], xp[2], 0.f};
        }
      }
      *reinterpret_cast<f32x4*>(&xs[pp][t][0]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < P / 4; ++q) {
      const int pp = pg * (P / 4) + q;
      const float g = dys[pp][col];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[pp][t][0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[t][c] = fmaf(g, xv[c], acc[t][c]);
      }
    }
    __syncthreads();
  }
  if (pg > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) red[pg - 1][col][t * 3 + c] = acc[t][c];
  }
  __syncthreads();
  if (pg == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = ((acc[t][c] + red[0][col][t * 3 + c]) + red[1][col][t * 3 + c]) + red[2][col][t * 3 + c];
        atomicAdd(dw + ((size_t)(co0 + col) * 9 + t) * 3 + c, v);
      }
  }
}

// dw[co][t][ci] += sum_pix dy[pix][co] * x[pix_t][ci]; thread = (co, tap group), block = pixel range
template <int CMAX>
__global__ __launch_bounds__(256) void conv_small_cin_wgrad_kernel(const float* __restrict__ x, int ldx,
                                                                   const float* __restrict__ dy, int lddy,
                                                                   float* __restrict__ dw, int B, int Hi,
                                                                   int Wi, int Cin, int Ho, int Wo, int Cout,
                                                                   int ks, int stride, int pix_per_block) {
  // Block = 64 output channels x a pixel range, walked in 64-pixel stages through LDS: all 256
  // threads stage the dy tile [64 pix][64 co] and the gathered input taps [64 pix][T][CMAX] with
  // independent loads (memory-level parallelism instead of a serial per-pixel loop), then thread
  // (co, tap group tg: taps tg, tg+4, tg+8) runs the 64-pixel FMA loop out of LDS (dy: conflict-free,
  // x: broadcast).  One atomic per weight per block; the grid keeps blocks x weights small.
  constexpr int P = 64;
  __shared__ __attribute__((aligned(16))) float dys[P][64];
  __shared__ __attribute__((aligned(16))) float xs[P][12][CMAX];
  const int T = ks * ks, pad = (ks - 1) / 2;
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * 64, co = co0 + (tid & 63);
  const int tg = tid >> 6;
  const int M = B * Ho * Wo, hw = Ho * Wo;
  const int lo = blockIdx.x * pix_per_block;
  const int hi = min(lo + pix_per_block, M);
  float acc[3][CMAX];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < CMAX; ++c) acc[a][c] = 0.f;
  for (int s0 = lo; s0 < hi; s0 += P) {
    // dy tile: thread -> (pixel j>>4, 4 channels (j&15)*4)
#pragma unroll
    for (int r = 0; r < (P * 16) / 256; ++r) {
      const int j = tid + 256 * r;
      const int pp = j >> 4, c4 = (j & 15) * 4;
      const int m = s0 + pp;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < hi) {
        const float* src = dy + (size_t)m * lddy + co0 + c4;
        if (co0 + c4 + 3 < Cout && (lddy & 3) == 0) v = *reinterpret_cast<const f32x4*>(src);
        else
          for (int e = 0; e < 4; ++e) if (co0 + c4 + e < Cout) v[e] = src[e];
      }
      *reinterpret_cast<f32x4*>(&dys[pp][c4]) = v;
    }
    // input taps: item -> (pixel, tap)
    for (int j = tid; j < P * T; j += 256) {
      const int pp = j / T, t = j - pp * T;
      const int m = s0 + pp;
      float v[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) v[c] = 0.f;
      if (m < hi) {
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const int iy = oy * stride + t / ks - pad, ix = ox * stride + t % ks - pad;
        if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) {
          const float* xp = x + ((size_t)(b * Hi + iy) * Wi + ix) * ldx;
#pragma unroll
          for (int c = 0; c < CMAX; ++c) if (c < Cin) v[c] = xp[c];
        }
      }
#pragma unroll
      for (int c = 0; c < CMAX; ++c) xs[pp][t][c] = v[c];
    }
    __syncthreads();
#pragma unroll 8
    for (int pp = 0; pp < P; ++pp) {
      const float g = dys[pp][tid & 63];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) acc[a][c] = fmaf(g, xs[pp][tg + 4 * a][c], acc[a][c]);
      }
    }
    __syncthreads();
  }
  if (co >= Cout) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int t = tg + 4 * a;
    if (t >= T) continue;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < Cin) atomicAdd(dw + ((size_t)co * T + t) * Cin + c, acc[a][c]);
  }
}

// dx[b,iy,ix,ci] (+)= sum_{t,co} dy[b,oy,ox,co] * w[co][t][ci] over the (t, oy, ox) with oy*s + kh - pad == iy (same for x):
// gather form, one thread per input pixel, the whole weight ([Cout][T][Cin], Cout <= 64 per pass) in LDS as [t][co][CMAX].
// Only the logit channels of a concatenated first-layer input need this gradient; it is cheap enough to write all Cin.
__global__ __launch_bounds__(256) void conv_small_cin_dgrad_kernel(const float* __restrict__ dy, int lddy,
                                                                   const float* __restrict__ w, float* __restrict__ dx,
                                                                   int lddx, int accumulate, int B, int Hi, int Wi, int Cin,
                                                                   int Ho, int Wo, int Cout, int ks, int stride) {
  constexpr int CMAX = HRSEG_SMALL_CIN_MAX;
  __shared__ __attribute__((aligned(16))) float wl[9 * 64 * CMAX];
  const int T = ks * ks, pad = (ks - 1) / 2;
  const long npix = (long)B * Hi * Wi;
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  int b = 0, iy = 0, ix = 0;
  if (m < npix) {
    b = (int)(m / ((long)Hi * Wi));
    const int rem = (int)(m - (long)b * Hi * Wi);
    iy = rem / Wi;
    ix = rem - iy * Wi;
  }
  float acc[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
  for (int co0 = 0; co0 < Cout; co0 += 64) {            // 64 output channels per pass through LDS
    __syncthreads();
    for (int i = threadIdx.x; i < T * 64 * CMAX; i += 256) {
      const int c = i % CMAX, co = (i / CMAX) % 64, t = i / (CMAX * 64);
      wl[i] = (co0 + co < Cout && c < Cin) ? w[((size_t)(co0 + co) * T + t) * Cin + c] : 0.f;
    }
    __syncthreads();
    if (m >= npix) continue;
    const int nco = min(64, Cout - co0);
    for (int t = 0; t < T; ++t) {
      const int ny = iy + pad - t / ks, nx = ix + pad - t % ks;       // = oy*s, ox*s
      if (ny < 0 || nx < 0 || (ny % stride) || (nx % stride)) continue;
      const int oy = ny / stride, ox = nx / stride;
      if (oy >= Ho || ox >= Wo) continue;
      const float* g = dy + ((size_t)(b * Ho + oy) * Wo + ox) * lddy + co0;
      const float* wt = wl + (size_t)t * 64 * CMAX;
      for (int co = 0; co < nco; co += 4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + co);     // Cout % 16 == 0, lddy % 4 == 0 (host-checked)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CMAX; ++c) acc[c] = fmaf(gv[e], wt[(co + e) * CMAX + c], acc[c]);
      }
    }
  }
  if (m >= npix) return;
  float* d = dx + (size_t)m * lddx;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < Cin) d[c] = accumulate ? d[c] + acc[c] : acc[c];
}

void launch_small_cin_fwd(const float* x, const float* w, const float* bias, float* y, const hrseg_conv_shape_t* s, hipStream_t st) {
  const float* res = s->residual;
  const int ldr = s->ldr, relu = s->relu;
  const long M = (long)s->B * s->Ho * s->Wo;
  if (s->Cin == 3 && s->ksize == 3 && s->Cout % 64 == 0 && s->ldy % 4 == 0 && (!res || ldr % 4 == 0) && M < (1L << 31) && g_small_cin3) {
    long gx = ceil_div(M, 16);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(conv_cin3_k3_fwd_kernel, dim3((unsigned)gx, s->Cout / 64), dim3(256), 0, st, x, s->ldx, w, bias, y, s->ldy, s->B,
                       s->Hi, s->Wi, s->Ho, s->Wo, s->stride, res, ldr, relu);
    return;
  }
  dim3 grid(ceil_div(M, 64), ceil_div(s->Cout, 64));
  if (s->Cin <= 4)
    hipLaunchKernelGGL(conv_small_cin_fwd_kernel<4>, grid, dim3(256), 0, st, x, s->ldx, w, bias, y, s->ldy, s->B, s->Hi, s->Wi,
                       s->Cin, s->Ho, s->Wo, s->Cout, s->ksize, s->stride, res, ldr, relu);
  else
    hipLaunchKernelGGL(conv_small_cin_fwd_kernel<8>, grid, dim3(256), 0, st, x, s->ldx, w, bias, y, s->ldy, s->B, s->Hi, s->Wi,
                       s->Cin, s->Ho, s->Wo, s->Cout, s->ksize, s->stride, res, ldr, relu);
}
void launch_small_cin_wgrad(const float* x, const float* dy, float* dw, const hrseg_conv_shape_t* s, int ppb, hipStream_t st) {
  const long M = (long)s->B * s->Ho * s->Wo;
  if (s->Cin == 3 && s->ksize == 3 && s->Cout % 64 == 0 && s->ldy % 4 == 0 && M < (1L << 31) && g_small_cin3) {
    hipLaunchKernelGGL(conv_cin3_k3_wgrad_kernel, dim3(ceil_div(M, ppb), s->Cout / 64), dim3(256), 0, st, x, s->ldx, dy, s->ldy, dw,
                       s->B, s->Hi, s->Wi, s->Ho, s->Wo, s->stride, ppb);
    return;
  }
  dim3 grid(ceil_div(M, ppb), ceil_div(s->Cout, 64));
  if (s->Cin <= 4)
    hipLaunchKernelGGL(conv_small_cin_wgrad_kernel<4>, grid, dim3(256), 0, st, x, s->ldx, dy, s->ldy, dw, s->B, s->Hi, s->Wi,
                       s->Cin, s->Ho, s->Wo, s->Cout, s->ksize, s->stride, ppb);
  else
    hipLaunchKernelGGL(conv_small_cin_wgrad_kernel<8>, grid, dim3(256), 0, st, x, s->ldx, dy, s->ldy, dw, s->B, s->Hi, s->Wi,
                       s->Cin, s->Ho, s->Wo, s->Cout, s->ksize, s->stride, ppb);
}
void launch_small_cin_dgrad(const float* dy, const float* w, float* dx, int accumulate, const hrseg_conv_shape_t* s, hipStream_t st) {
  const long npix = (long)s->B * s->Hi * s->Wi;
  hipLaunchKernelGGL(conv_small_cin_dgrad_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, st, dy, s->ldy, w, dx, s->ldx,
                     accumulate, s->B, s->Hi, s->Wi, s->Cin, s->Ho, s->Wo, s->Cout, s->ksize, s->stride);
}
