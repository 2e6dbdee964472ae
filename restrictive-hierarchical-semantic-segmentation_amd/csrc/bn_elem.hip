// Batch-norm (training statistics, apply, backward), pooling, bilinear resize and
// elementwise glue on NHWC fp32 for gfx950.  All HBM-bound: one pass per tensor,
// 16-byte accesses, per-channel parameters held in registers.
//
// Thread mapping used throughout ("pixel lanes x channel quads"): with
// Q = C/4 channel quads, thread t owns quad t % Q for pixel lane t / Q
// (P = 256/Q lanes per block; threads beyond P*Q idle).  Consecutive threads
// read consecutive 16-byte pieces of a pixel row, so a wave covers whole rows.
#include "common.h"

struct Lanes {
  int Q, P, cq, pl;
  bool active;
};
__device__ __forceinline__ Lanes make_lanes(int C) {
  Lanes l;
  l.Q = C >> 2;
  l.P = (l.Q >= 256) ? 1 : 256 / l.Q;
  l.cq = threadIdx.x % l.Q;
  l.pl = threadIdx.x / l.Q;
  l.active = l.pl < l.P;
  return l;
}
static int elem_grid(long npix, int C) {
  const int Q = C / 4, P = (Q >= 256) ? 1 : 256 / Q;
  long blocks = (npix + P - 1) / P;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
#define FOR_PIXELS(pix, L, npix) \
  for (long pix = (long)blockIdx.x * (L).P + (L).pl; pix < (npix); pix += (long)gridDim.x * (L).P)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// y*scale + shift with ONE rounding per element: the forward pass and the backward pass's recomputed
// ReLU mask must evaluate the identical expression
__device__ __forceinline__ f32x4 bn_affine(f32x4 y, f32x4 sc, f32x4 sh) {
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(y[j], sc[j], sh[j]);
  return v;
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// C > 1024 is not supported by the quad mapping (Q must be <= 256): the widest
// tensor on the path is the UNet 1024-channel concat.
static int check_c(int C, const char* who) {
  HRSEG_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 1024, "%s: C=%d must be a multiple of 4 and <= 1024", who, C);
  return 0;
}

// --------------------------------------------------------------------------- BN statistics
// partial[chunk][0][c] = sum y, partial[chunk][1][c] = sum y*y over the chunk's pixels
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, int ldy, long npix, int C,
                                                       double* __restrict__ partial, long pix_per_chunk) {
  __shared__ double red[256 * 8];
  const Lanes L = make_lanes(C);
  const long lo = (long)blockIdx.x * pix_per_chunk;
  const long hi = (lo + pix_per_chunk < npix) ? lo + pix_per_chunk : npix;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
  if (L.active)
    for (long pix = lo + L.pl; pix < hi; pix += L.P) {
      const f32x4 v = ld4(y + pix * ldy + 4 * L.cq);
      s += v;
      ss += v * v;
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[threadIdx.x * 8 + j] = s[j];
    red[threadIdx.x * 8 + 4 + j] = ss[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int pl = 0; pl < L.P; ++pl) {
      const int t = pl * L.Q + (c >> 2);
      a += red[t * 8 + (c & 3)];
      b += red[t * 8 + 4 + (c & 3)];
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

// Sum of partial[k][which][c] over chunks k for the block's 16 channels: 16 chunk lanes per
// channel (a serial loop over up to 1024 chunks is a chain of dependent HBM-latency loads).
// Returns the totals to the threads with lane==0 (tid < 16).
__device__ __forceinline__ void reduce_chunks16(const double* __restrict__ partial, int nchunks, int C, int c,
                                               double& s0, double& s1, double* red) {
  const int lane = threadIdx.x >> 4;  // 0..15
  double a = 0.0, b = 0.0;
  if (c < C) {
    // up to 16 chunks per lane: issue the loads in batches of 4 chunks (8 independent loads in flight)
    // instead of a load-add chain of cold-memory latencies
    int k = lane;
    for (; k + 48 < nchunks; k += 64) {
      double t0[4], t1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        t0[u] = partial[((size_t)(k + 16 * u) * 2 + 0) * C + c];
        t1[u] = partial[((size_t)(k + 16 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a += t0[u];
        b += t1[u];
      }
    }
    for (; k < nchunks; k += 16) {
      a += partial[((size_t)k * 2 + 0) * C + c];
      b += partial[((size_t)k * 2 + 1) * C + c];
    }
  }
  red[threadIdx.x * 2] = a;
  red[threadIdx.x * 2 + 1] = b;
  __syncthreads();
  s0 = 0.0;
  s1 = 0.0;
  if (threadIdx.x < 16)
    for (int l = 0; l < 16; ++l) {
      s0 += red[(l * 16 + threadIdx.x) * 2];
      s1 += red[(l * 16 + threadIdx.x) * 2 + 1];
    }
}

// coef: [0]=mean [1]=rstd [2]=scale [3]=shift.  grid = ceil(C/16) blocks of 256 threads.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partial, int nchunks, long npix,
                                                          int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, long long* nbt, float momentum,
                                                          float eps, float* __restrict__ coef) {
  __shared__ double red[512];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  double s, ss;
  reduce_chunks16(partial, nchunks, C, c, s, ss, red);
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (threadIdx.x >= 16 || c >= C) return;
  const double mean = s / (double)npix;
  double var = ss / (double)npix - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  coef[c] = (float)mean;
  coef[C + c] = rstd;
  coef[2 * C + c] = g * rstd;
  coef[3 * C + c] = b - (float)mean * g * rstd;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) {
    const double unb = (npix > 1) ? var * (double)npix / (double)(npix - 1) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// BatchNorm (running statistics) folded into the preceding convolution's weights and bias: block = one output channel
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                                      int row, float* __restrict__ w_out, float* __restrict__ b_out) {
  const int co = blockIdx.x;
  const float s = (gamma ? gamma[co] : 1.f) / sqrtf(rvar[co] + eps);
  for (int i = threadIdx.x; i < row; i += 256) w_out[(size_t)co * row + i] = w[(size_t)co * row + i] * s;
  if (threadIdx.x == 0) b_out[co] = ((bias ? bias[co] : 0.f) - rmean[co]) * s + (beta ? beta[co] : 0.f);
}

__global__ void bn_eval_coef_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                    float eps, int C, float* coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(rvar[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  coef[c] = rmean[c];
  coef[C + c] = rstd;
  coef[2 * C + c] = g * rstd;
  coef[3 * C + c] = b - rmean[c] * g * rstd;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int ldy,
                                                       const float* __restrict__ coef,
                                                       const float* __restrict__ res, int ldr, int relu,
                                                       float* __restrict__ z, int ldz, long npix, int C) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  const f32x4 sc = ld4(coef + 2 * C + 4 * L.cq), sh = ld4(coef + 3 * C + 4 * L.cq);
  FOR_PIXELS(pix, L, npix) {
    f32x4 v = ld4(y + pix * ldy + 4 * L.cq) * sc + sh;
    if (res) v += ld4(res + pix * ldr + 4 * L.cq);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    st4(z + pix * ldz + 4 * L.cq, v);
  }
}

// partial[chunk][0][c] = sum g, [1][c] = sum g*xhat, with g = dz * (z > 0 if relu)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, int lddz,
                                                            const float* __restrict__ z, int ldz, int relu,
                                                            const float* __restrict__ y, int ldy,
                                                            const float* __restrict__ coef, long npix, int C,
                                                            double* __restrict__ partial, long pix_per_chunk) {
  __shared__ double red[256 * 8];
  const Lanes L = make_lanes(C);
  const long lo = (long)blockIdx.x * pix_per_chunk;
  const long hi = (lo + pix_per_chunk < npix) ? lo + pix_per_chunk : npix;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, sx = {0.f, 0.f, 0.f, 0.f};
  if (L.active) {
    const f32x4 mean = ld4(coef + 4 * L.cq), rstd = ld4(coef + C + 4 * L.cq);
    for (long pix = lo + L.pl; pix < hi; pix += L.P) {
      f32x4 g = ld4(dz + pix * lddz + 4 * L.cq);
      if (relu) {
        const f32x4 zz = ld4(z + pix * ldz + 4 * L.cq);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = zz[j] > 0.f ? g[j] : 0.f;
      }
      const f32x4 xh = (ld4(y + pix * ldy + 4 * L.cq) - mean) * rstd;
      s += g;
      sx += g * xh;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[threadIdx.x * 8 + j] = s[j];
    red[threadIdx.x * 8 + 4 + j] = sx[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int pl = 0; pl < L.P; ++pl) {
      const int t = pl * L.Q + (c >> 2);
      a += red[t * 8 + (c & 3)];
      b += red[t * 8 + 4 + (c & 3)];
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

// totals -> totals[0..1][c] (a separate [2][C] area behind the partials), dgamma/dbeta accumulation
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial,
                                                              double* __restrict__ totals, int nchunks, int C,
                                                              float* dgamma, float* dbeta) {
  __shared__ double red[512];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  double s, sx;
  reduce_chunks16(partial, nchunks, C, c, s, sx, red);
  if (threadIdx.x >= 16 || c >= C) return;
  totals[c] = s;
  totals[C + c] = sx;
  if (dgamma) dgamma[c] += (float)sx;
  if (dbeta) dbeta[c] += (float)s;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const double* __restrict__ totals,
                                                           const float* __restrict__ dz, int lddz,
                                                           const float* __restrict__ z, int ldz, int relu,
                                                           const float* __restrict__ y, int ldy,
                                                           const float* __restrict__ coef,
                                                           float* __restrict__ dy, int lddy,
                                                           float* __restrict__ dres, int lddres, int dres_acc,
                                                           long npix, int C, int eval_mode) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  const f32x4 mean = ld4(coef + 4 * L.cq), rstd = ld4(coef + C + 4 * L.cq), scale = ld4(coef + 2 * C + 4 * L.cq);
  f32x4 mg, mgx;
  const float inv = eval_mode ? 0.f : (float)(1.0 / (double)npix);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mg[j] = (float)(totals[4 * L.cq + j]) * inv;
    mgx[j] = (float)(totals[C + 4 * L.cq + j]) * inv;
  }
  FOR_PIXELS(pix, L, npix) {
    f32x4 g = ld4(dz + pix * lddz + 4 * L.cq);
    if (relu) {
      const f32x4 zz = ld4(z + pix * ldz + 4 * L.cq);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = zz[j] > 0.f ? g[j] : 0.f;
    }
    const f32x4 xh = (ld4(y + pix * ldy + 4 * L.cq) - mean) * rstd;
    st4(dy + pix * lddy + 4 * L.cq, scale * (g - mg - xh * mgx));
    if (dres) {
      float* d = dres + pix * lddres + 4 * L.cq;
      st4(d, dres_acc ? ld4(d) + g : g);
    }
  }
}

// --------------------------------------------------------------------------- max pool 2x2 (floor)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, int ldx,
                                                           float* __restrict__ y, int ldy, int B, int Hi, int Wi,
                                                           int C) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  const int Ho = Hi / 2, Wo = Wi / 2;
  const long npix = (long)B * Ho * Wo;
  FOR_PIXELS(pix, L, npix) {
    const int b = (int)(pix / ((long)Ho * Wo));
    const int rem = (int)(pix - (long)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const float* p = x + (((size_t)b * Hi + 2 * oy) * Wi + 2 * ox) * ldx + 4 * L.cq;
    f32x4 m = ld4(p);
    const f32x4 v1 = ld4(p + ldx), v2 = ld4(p + (size_t)Wi * ldx), v3 = ld4(p + (size_t)Wi * ldx + ldx);
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(m[j], v1[j]), fmaxf(v2[j], v3[j]));
    st4(y + pix * ldy + 4 * L.cq, m);
  }
}

// gradient goes to the first maximum in scan order (strict >), as torch's max_pool2d
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ dy, int lddy,
                                                           float* __restrict__ dx, int lddx, int acc, int B,
                                                           int Hi, int Wi, int C) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  const int Hc = (Hi + 1) / 2, Wc = (Wi + 1) / 2;  // cells incl. the odd leftover row/col
  const int Ho = Hi / 2, Wo = Wi / 2;
  const long ncell = (long)B * Hc * Wc;
  FOR_PIXELS(cell, L, ncell) {
    const int b = (int)(cell / ((long)Hc * Wc));
    const int rem = (int)(cell - (long)b * Hc * Wc);
    const int oy = rem / Wc, ox = rem - oy * Wc;
    const size_t base = (((size_t)b * Hi + 2 * oy) * Wi + 2 * ox);
    const bool full = (oy < Ho) & (ox < Wo);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    int arg[4] = {-1, -1, -1, -1};
    if (full) {
      g = ld4(dy + (((size_t)b * Ho + oy) * Wo + ox) * lddy + 4 * L.cq);
      f32x4 m = ld4(x + base * ldx + 4 * L.cq);
#pragma unroll
      for (int j = 0; j < 4; ++j) arg[j] = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const f32x4 v = ld4(x + (base + (k >> 1) * Wi + (k & 1)) * ldx + 4 * L.cq);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > m[j]) { m[j] = v[j]; arg[j] = k; }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
      if (iy >= Hi || ix >= Wi) continue;
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (arg[j] == k) ? g[j] : 0.f;
      float* d = dx + (base + (k >> 1) * Wi + (k & 1)) * lddx + 4 * L.cq;
      st4(d, acc ? ld4(d) + o : o);
    }
  }
}

// --------------------------------------------------------------------------- bilinear
// source coordinate exactly as torch's upsample_bilinear2d (fp32 arithmetic)
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int align, int& i0, int& i1,
                                          float& l0, float& l1) {
  float r;
  if (align) {
    r = scale * (float)o;
  } else {
    r = scale * ((float)o + 0.5f) - 0.5f;
    if (r < 0.f) r = 0.f;
  }
  i0 = (int)r;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = r - (float)i0;
  l0 = 1.f - l1;
}
static float resize_scale(int in_size, int out_size, int align) {
  if (align) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  return (float)in_size / (float)out_size;
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ in, int ldin, int B, int Hi,
                                                           int Wi, int C, float* __restrict__ out, int ldout,
                                                           int Hout, int Wout, int Hr, int Wr, int py, int px,
                                                           float sh, float sw, int align, int acc, int relu) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  const long npix = (long)B * Hout * Wout;
  FOR_PIXELS(pix, L, npix) {
    const int b = (int)(pix / ((long)Hout * Wout));
    const int rem = (int)(pix - (long)b * Hout * Wout);
    const int oy = rem / Wout - py, ox = rem % Wout - px;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (oy >= 0 && oy < Hr && ox >= 0 && ox < Wr) {
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      src_index(oy, sh, Hi, align, y0, y1, ly0, ly1);
      src_index(ox, sw, Wi, align, x0, x1, lx0, lx1);
      const float* p = in + (size_t)b * Hi * Wi * ldin + 4 * L.cq;
      const f32x4 v00 = ld4(p + ((size_t)y0 * Wi + x0) * ldin), v01 = ld4(p + ((size_t)y0 * Wi + x1) * ldin);
      const f32x4 v10 = ld4(p + ((size_t)y1 * Wi + x0) * ldin), v11 = ld4(p + ((size_t)y1 * Wi + x1) * ldin);
      v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    }
    float* o = out + pix * ldout + 4 * L.cq;
    if (acc) v += ld4(o);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    st4(o, v);
  }
}

// HRNet fuse sum (Models/models.py:527-542) in ONE pass: out = relu?( sum of same-resolution terms + sum of bilinearly
// up-sampled low-resolution terms ).  The chain of add / copy / accumulating-bilinear launches it replaces re-reads and
// re-writes `out` once per term.
struct FuseSumArgs {
  int n_same, n_low;
  const float* same[4]; int ld_same[4];
  const float* low[3]; int ld_low[3]; int Hi[3], Wi[3];
  float sh[3], sw[3];
  float* out; int ldo;
  int B, H, W, C, align, relu;
};
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseSumArgs a) {
  const Lanes L = make_lanes(a.C);
  if (!L.active) return;
  const long npix = (long)a.B * a.H * a.W;
  FOR_PIXELS(pix, L, npix) {
    f32x4 v = ld4(a.same[0] + pix * a.ld_same[0] + 4 * L.cq);
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < a.n_same) v += ld4(a.same[i] + pix * a.ld_same[i] + 4 * L.cq);
    if (a.n_low) {
      const int b = (int)(pix / ((long)a.H * a.W));
      const int rem = (int)(pix - (long)b * a.H * a.W);
      const int oy = rem / a.W, ox = rem - oy * a.W;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < a.n_low) {
          int y0, y1, x0, x1;
          float ly0, ly1, lx0, lx1;
          src_index(oy, a.sh[j], a.Hi[j], a.align, y0, y1, ly0, ly1);
          src_index(ox, a.sw[j], a.Wi[j], a.align, x0, x1, lx0, lx1);
          const int ld = a.ld_low[j], Wi = a.Wi[j];
          const float* p = a.low[j] + (size_t)b * a.Hi[j] * Wi * ld + 4 * L.cq;
          const f32x4 v00 = ld4(p + ((size_t)y0 * Wi + x0) * ld), v01 = ld4(p + ((size_t)y0 * Wi + x1) * ld);
          const f32x4 v10 = ld4(p + ((size_t)y1 * Wi + x0) * ld), v11 = ld4(p + ((size_t)y1 * Wi + x1) * ld);
          v += ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        }
    }
    if (a.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    st4(a.out + pix * a.ldo + 4 * L.cq, v);
  }
}

// gather form of the transpose: every input pixel sums the output pixels that read it (no atomics,
// deterministic).  An 8x resize gives each input pixel an ~18x18 footprint and only a few thousand input
// pixels: RS sub-lanes per pixel split the footprint rows (thread = channel quad x pixel lane x row split)
// and reduce through LDS, so the small tensors still fill the machine.
template <int RS>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dout, int lddout, int B,
                                                           int Hi, int Wi, int C, float* __restrict__ din,
                                                           int lddin, int Hout, int Wout, int Hr, int Wr, int py,
                                                           int px, float sh, float sw, int align, int acc) {
  __shared__ f32x4 red[256];
  const int Q = C >> 2;
  const int P2 = max(1, 256 / (Q * RS));           // pixel lanes per block
  const int cq = threadIdx.x % Q, pl = (threadIdx.x / Q) % P2, rs = threadIdx.x / (Q * P2);
  const bool active = rs < RS && (int)threadIdx.x < Q * P2 * RS;
  const long npix = (long)B * Hi * Wi;
  const float ish = sh > 0.f ? 1.f / sh : 0.f, isw = sw > 0.f ? 1.f / sw : 0.f;
  for (long base = (long)blockIdx.x * P2; base < npix; base += (long)gridDim.x * P2) {
    const long pix = base + pl;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (active && pix < npix) {
      const int b = (int)(pix / ((long)Hi * Wi));
      const int rem = (int)(pix - (long)b * Hi * Wi);
      const int iy = rem / Wi, ix = rem - iy * Wi;
      // candidate output rows/cols: src in (iy-1, iy+1)  (whole range when scale is 0)
      int oy_lo = 0, oy_hi = Hr - 1, ox_lo = 0, ox_hi = Wr - 1;
      if (sh > 0.f) {
        oy_lo = max(0, (int)floorf(((float)iy - 1.f + (align ? 0.f : 0.5f)) * ish - (align ? 0.f : 0.5f)) - 1);
        oy_hi = min(Hr - 1, (int)ceilf(((float)iy + 1.f + (align ? 0.f : 0.5f)) * ish - (align ? 0.f : 0.5f)) + 1);
      }
      if (sw > 0.f) {
        ox_lo = max(0, (int)floorf(((float)ix - 1.f + (align ? 0.f : 0.5f)) * isw - (align ? 0.f : 0.5f)) - 1);
        ox_hi = min(Wr - 1, (int)ceilf(((float)ix + 1.f + (align ? 0.f : 0.5f)) * isw - (align ? 0.f : 0.5f)) + 1);
      }
      for (int oy = oy_lo + rs; oy <= oy_hi; oy += RS) {
        int y0, y1;
        float ly0, ly1;
        src_index(oy, sh, Hi, align, y0, y1, ly0, ly1);
        const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
        if (wy == 0.f) continue;
        const float* row = dout + (((size_t)b * Hout + oy + py) * Wout + px) * lddout + 4 * cq;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
          int x0, x1;
          float lx0, lx1;
          src_index(ox, sw, Wi, align, x0, x1, lx0, lx1);
          const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
          if (wx == 0.f) continue;
          s += (wy * wx) * ld4(row + (size_t)ox * lddout);
        }
      }
    }
    if (RS > 1) {
      __syncthreads();
      red[threadIdx.x] = s;
      __syncthreads();
      if (active && rs == 0) {
#pragma unroll
        for (int r = 1; r < RS; ++r) s += red[threadIdx.x + r * Q * P2];
      }
    }
    if (active && rs == 0 && pix < npix) {
      float* d = din + pix * lddin + 4 * cq;
      st4(d, acc ? ld4(d) + s : s);
    }
  }
}

// --------------------------------------------------------------------------- small glue
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                  int ldb, float* __restrict__ out, int ldo, int relu, long npix,
                                                  int C) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  FOR_PIXELS(pix, L, npix) {
    f32x4 v = ld4(a + pix * lda + 4 * L.cq) + ld4(b + pix * ldb + 4 * L.cq);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    st4(out + pix * ldo + 4 * L.cq, v);
  }
}
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ in, int ldin, float* __restrict__ out,
                                                   int ldout, int acc, long npix, int C) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  FOR_PIXELS(pix, L, npix) {
    f32x4 v = ld4(in + pix * ldin + 4 * L.cq);
    float* o = out + pix * ldout + 4 * L.cq;
    st4(o, acc ? ld4(o) + v : v);
  }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dz, int lddz,
                                                       const float* __restrict__ z, int ldz, float* __restrict__ dx,
                                                       int lddx, long npix, int C) {
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  FOR_PIXELS(pix, L, npix) {
    f32x4 g = ld4(dz + pix * lddz + 4 * L.cq);
    const f32x4 zz = ld4(z + pix * ldz + 4 * L.cq);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = zz[j] > 0.f ? g[j] : 0.f;
    st4(dx + pix * lddx + 4 * L.cq, g);
  }
}

// max|x| of an NHWC tensor into a 64-slot array (slot = block % 64, as hrseg_bn_bwd_t.dy_absmax); NaN counts as +Inf so
// that a non-finite tensor is never reported as in range.  `out` must be zeroed by the caller (hrseg_fill).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int ldx, long npix, int C,
                                                     float* __restrict__ out) {
  const Lanes L = make_lanes(C);
  float amax = 0.f;
  if (L.active) {
    FOR_PIXELS(pix, L, npix) {
      const f32x4 v = ld4(x + pix * ldx + 4 * L.cq);
#pragma unroll
      for (int j = 0; j < 4; ++j) amax = fmaxf(amax, (v[j] != v[j]) ? __builtin_inff() : fabsf(v[j]));
    }
  }
  __shared__ float wmax[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(out) + (blockIdx.x & 63), __float_as_uint(m));
  }
}

// NCHW <-> NHWC for narrow tensors (image: C=3, logits: C<=16): thread per pixel
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int ldout, int B, int C,
                                    long hw) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * hw) return;
  const long b = i / hw, p = i - b * hw;
  for (int c = 0; c < C; ++c) out[i * ldout + c] = in[(b * C + c) * hw + p];
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int ldin, float* __restrict__ out, int B, int C,
                                    long hw) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * hw) return;
  const long b = i / hw, p = i - b * hw;
  for (int c = 0; c < C; ++c) out[(b * C + c) * hw + p] = in[i * ldin + c];
}

__global__ void fill_kernel(float* p, float v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// AdamW over one flat buffer (torch.optim.AdamW single-tensor semantics)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n4, long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float rsqrt_bc2, float gscale) {
  const float step = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = ld4(p + 4 * i), gg = ld4(g + 4 * i) * gscale, mm = ld4(m + 4 * i), vv = ld4(v + 4 * i);
    pp = pp * (1.f - lr * wd);
    mm = mm + (gg - mm) * (1.f - b1);          // lerp, as torch: m.lerp_(g, 1-b1)
    vv = vv * b2 + gg * gg * (1.f - b2);
#pragma unroll
    for (int j = 0; j < 4; ++j) pp[j] -= step * mm[j] / (sqrtf(vv[j]) * rsqrt_bc2 + eps);
    st4(p + 4 * i, pp);
    st4(m + 4 * i, mm);
    st4(v + 4 * i, vv);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - 4 * n4)) {  // tail
    const long i = 4 * n4 + threadIdx.x;
    float pp = p[i] * (1.f - lr * wd), gg = g[i] * gscale;
    const float mm = m[i] + (gg - m[i]) * (1.f - b1), vv = v[i] * b2 + gg * gg * (1.f - b2);
    p[i] = pp - step * mm / (sqrtf(vv) * rsqrt_bc2 + eps);
    m[i] = mm;
    v[i] = vv;
  }
}

// Graph-replayable AdamW: step count and hyper-parameters live in device memory, so a captured
// launch picks up the next step's bias correction and a scheduler's new lr on every replay.
// hyper = {lr, beta1, beta2, eps, weight_decay, grad_scale}; state = {step, bc1, 1/sqrt(bc2)}
__global__ void adam_tick_kernel(float* state, const float* hyper) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double step = (double)state[0] + 1.0;
    state[0] = (float)step;
    state[1] = (float)(1.0 - pow((double)hyper[1], step));
    state[2] = (float)(1.0 / sqrt(1.0 - pow((double)hyper[2], step)));
  }
}
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n4, long n,
                                                        const float* __restrict__ hyper,
                                                        const float* __restrict__ state) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], gscale = hyper[5];
  const float step = lr / state[1], rsqrt_bc2 = state[2];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = ld4(p + 4 * i), gg = ld4(g + 4 * i) * gscale, mm = ld4(m + 4 * i), vv = ld4(v + 4 * i);
    pp = pp * (1.f - lr * wd);
    mm = mm + (gg - mm) * (1.f - b1);
    vv = vv * b2 + gg * gg * (1.f - b2);
#pragma unroll
    for (int j = 0; j < 4; ++j) pp[j] -= step * mm[j] / (sqrtf(vv[j]) * rsqrt_bc2 + eps);
    st4(p + 4 * i, pp);
    st4(m + 4 * i, mm);
    st4(v + 4 * i, vv);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - 4 * n4)) {
    const long i = 4 * n4 + threadIdx.x;
    float pp = p[i] * (1.f - lr * wd), gg = g[i] * gscale;
    const float mm = m[i] + (gg - m[i]) * (1.f - b1), vv = v[i] * b2 + gg * gg * (1.f - b2);
    p[i] = pp - step * mm / (sqrtf(vv) * rsqrt_bc2 + eps);
    m[i] = mm;
    v[i] = vv;
  }
}

// =========================================================================== grouped batch norm
// n independent BatchNorm problems (the parallel HRNet branches, or just one) per launch: block
// ranges [blk_end[g-1], blk_end[g]) belong to problem g.  Three launches forward (statistics,
// finalize, apply) and three backward (reduce, finalize, apply) whatever n is.  (Folding the finalize
// into the statistics kernel was measured slower three times -- last block reducing cold partials: 2-3x; fp64
// atomics into one accumulator: blocks finishing together serialise on 2*C addresses; sixteen accumulator rows and
// a last block that only adds those up (bn_last_block below, kept as an opt-in): +0.8 ms per step -- see DESIGN.md.)
#define BN_MAXG 8
struct BnGroupHdr { int n; int blk_end[BN_MAXG]; };
__device__ __forceinline__ int bn_find(const BnGroupHdr& h, int& local, int& nblk) {
  int g = 0;
  while (g + 1 < h.n && (int)blockIdx.x >= h.blk_end[g]) ++g;
  const int lo = g ? h.blk_end[g - 1] : 0;
  local = blockIdx.x - lo;
  nblk = h.blk_end[g] - lo;
  return g;
}
struct BnFwdG { BnGroupHdr h; hrseg_bn_fwd_t p[BN_MAXG]; };
struct BnBwdG { BnGroupHdr h; hrseg_bn_bwd_t p[BN_MAXG]; };

template <bool MASK>
__device__ __forceinline__ void stats_body(const float* __restrict__ a0, int ld0, const float* __restrict__ zmask,
                                           int ldz, int relu, const float* __restrict__ yy, int ldy,
                                           const float* __restrict__ coef, long npix, int C,
                                           double* __restrict__ partial, int chunk, int nchunks, bool bwd,
                                           double* red, int nseg = 1, const unsigned char* __restrict__ bmask = nullptr) {
  // forward: sums of a0 and a0^2; backward: sums of g and g*xhat with g = a0 * (zmask > 0 if relu)
  // nseg > 1: the pixel range is nseg equal segments (the batched level passes); a chunk never
  // straddles two segments (nchunks is a multiple of nseg)
  const Lanes L = make_lanes(C);
  const int cps = nchunks / nseg, seg = chunk / cps;
  const long seg_pix = npix / nseg;
  const long per = (seg_pix + cps - 1) / cps;
  const long lo = seg * seg_pix + (long)(chunk - seg * cps) * per;
  const long seg_end = (seg + 1) * seg_pix;
  const long hi = (lo + per < seg_end) ? lo + per : seg_end;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (L.active) {
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, rstd = {1.f, 1.f, 1.f, 1.f}, sc = mean, sh = mean;
    if (bwd) {
      mean = ld4(coef + 4 * L.cq);
      rstd = ld4(coef + C + 4 * L.cq);
      sc = ld4(coef + 2 * C + 4 * L.cq);
      sh = ld4(coef + 3 * C + 4 * L.cq);
    }
    // four pixels per trip, all loads issued before the first use: a chunk is walked by few threads, so the
    // bytes in flight per CU (what HBM-bound code lives on) come from this unrolling
    constexpr int U = 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long pix0 = lo + L.pl; pix0 < hi; pix0 += (long)U * L.P) {
      f32x4 v[U], yv[U], zz[U];      // (zz[u][0] carries the mask byte when the layer has one)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long pix = pix0 + (long)u * L.P;
        const bool ok = pix < hi;
        v[u] = ok ? ld4(a0 + pix * ld0 + 4 * L.cq) : zero;
        if (bwd) {
          yv[u] = ok ? ld4(yy + pix * ldy + 4 * L.cq) : mean;
          if (MASK && relu && bmask) zz[u][0] = __uint_as_float(ok ? (unsigned)bmask[pix * L.Q + L.cq] : 0u);
          else if (relu && zmask) zz[u] = ok ? ld4(zmask + pix * ldz + 4 * L.cq) : zero;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!bwd) {
          s += v[u];
          s2 += v[u] * v[u];
        } else {
          f32x4 gv = v[u];
          if (MASK && relu && bmask) {
            const unsigned bm = __float_as_uint(zz[u][0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = ((bm >> j) & 1u) ? gv[j] : 0.f;
          } else if (relu) {
            // z not given: the forward had no residual, so z > 0 <=> y*scale+shift > 0 (4 bytes less per element)
            const f32x4 zc = zmask ? zz[u] : bn_affine(yv[u], sc, sh);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = zc[j] > 0.f ? gv[j] : 0.f;
          }
          const f32x4 xh = (yv[u] - mean) * rstd;
          s += gv;
          s2 += gv * xh;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[threadIdx.x * 8 + j] = s[j];
    red[threadIdx.x * 8 + 4 + j] = s2[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int pl = 0; pl < L.P; ++pl) {
      const int t = pl * L.Q + (c >> 2);
      a += red[t * 8 + (c & 3)];
      b += red[t * 8 + 4 + (c & 3)];
    }
    partial[((size_t)chunk * 2 + 0) * C + c] = a;
    partial[((size_t)chunk * 2 + 1) * C + c] = b;
  }
}

__device__ __forceinline__ void bn_finalize_channel(const hrseg_bn_fwd_t& p, int c, double s, double ss) {
  const int C = p.C;
  // stat_updates > 1: the same batch statistics enter the running averages that many times (the level
  // passes of the hierarchical models run as one) -- sequential updates, bit for bit
  const int reps = p.stat_updates > 1 ? p.stat_updates : 1;
  const long npix = p.npix * (p.stat_ranks > 1 ? p.stat_ranks : 1);      // cross-rank statistics: sums over all ranks' pixels
  const double mean = s / (double)npix;
  double var = ss / (double)npix - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
  p.coef[c] = (float)mean;
  p.coef[C + c] = rstd;
  p.coef[2 * C + c] = ga * rstd;
  p.coef[3 * C + c] = be - (float)mean * ga * rstd;
  if (p.running_mean) {
    float rm = p.running_mean[c];
    for (int r = 0; r < reps; ++r) rm = (1.f - p.momentum) * rm + p.momentum * (float)mean;
    p.running_mean[c] = rm;
  }
  if (p.running_var) {
    // stat_div > 1: the tensor holds stat_div identical copies of the pass's images (batched level
    // passes); mean and biased variance of the copies are those of one pass, the unbiased factor uses
    // one pass's pixel count
    const long n1 = npix / (p.stat_div > 1 ? p.stat_div : 1);
    const double unb = (n1 > 1) ? var * (double)n1 / (double)(n1 - 1) : var;
    float rv = p.running_var[c];
    for (int r = 0; r < reps; ++r) rv = (1.f - p.momentum) * rv + p.momentum * (float)unb;
    p.running_var[c] = rv;
  }
}

__global__ __launch_bounds__(256) void bn_stats_group_kernel(BnFwdG g) {
  __shared__ double red[256 * 8];
  int local, nblk;
  const hrseg_bn_fwd_t& p = g.p[bn_find(g.h, local, nblk)];
  stats_body<false>(p.y, p.ldy, nullptr, 0, 0, nullptr, 0, nullptr, p.npix, p.C, p.partial, local, p.nchunks, false, red, 1);
}

__global__ __launch_bounds__(256) void bn_finalize_group_kernel(BnFwdG g) {
  __shared__ double red[512];
  int local, nblk;
  const hrseg_bn_fwd_t& p = g.p[bn_find(g.h, local, nblk)];
  const int C = p.C, c = local * 16 + (threadIdx.x & 15);
  double s, ss;
  reduce_chunks16(p.partial, p.nchunks, C, c, s, ss, red);
  if (local == 0 && threadIdx.x == 0 && p.num_batches_tracked)
    *(long long*)p.num_batches_tracked += (p.stat_updates > 1 ? p.stat_updates : 1);
  if (threadIdx.x >= 16 || c >= C) return;
  bn_finalize_channel(p, c, s, ss);
}

__global__ void bn_eval_coef_group_kernel(BnFwdG g) {
  int local, nblk;
  const hrseg_bn_fwd_t& p = g.p[bn_find(g.h, local, nblk)];
  const int c = local * 64 + threadIdx.x, C = p.C;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(p.running_var[c] + p.eps);
  const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
  p.coef[c] = p.running_mean[c];
  p.coef[C + c] = rstd;
  p.coef[2 * C + c] = ga * rstd;
  p.coef[3 * C + c] = be - p.running_mean[c] * ga * rstd;
}

__global__ __launch_bounds__(256) void bn_apply_group_kernel(BnFwdG g) {
  int local, nblk;
  const hrseg_bn_fwd_t& p = g.p[bn_find(g.h, local, nblk)];
  const int C = p.C;
  const Lanes L = make_lanes(C);
  if (!L.active) return;
  const f32x4 sc = ld4(p.coef + 2 * C + 4 * L.cq), sh = ld4(p.coef + 3 * C + 4 * L.cq);
  for (long pix = (long)local * L.P + L.pl; pix < p.npix; pix += (long)nblk * L.P) {
    f32x4 v = bn_affine(ld4(p.y + pix * p.ldy + 4 * L.cq), sc, sh);
    if (p.residual) {       // (residual_split: the block input is stored pre-split for its convolution readers; hi + lo is its value)
      const float* r = p.residual + pix * p.ldr + 4 * L.cq;
      v += p.residual_split ? hrseg_join_f16x2(*reinterpret_cast<const hrseg_u32x4*>(r)) : ld4(r);
    }
    if (p.relu) {
      // one byte per (pixel, channel quad): bit j = "channel 4q+j passed the ReLU".  With a residual the backward cannot
      // recompute the mask from y alone; reading this byte instead of z saves it 4 B per element, twice
      if (p.relu_mask) p.relu_mask[pix * L.Q + L.cq] = (unsigned char)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) | ((v[3] > 0.f) << 3));
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    // z_split: the tensor's only readers are fp16x2 convolutions that take their pixel operand pre-split (hrseg_conv_shape_t.
    // x_split): the granule goes out as {hi01, hi23, lo01, lo23} -- the same 16 bytes per 4 channels, the split done once here
    // (a bandwidth-bound kernel with VALU to spare) instead of by every staging wave of the readers
    if (p.z_split) *reinterpret_cast<hrseg_u32x4*>(p.z + pix * p.ldz + 4 * L.cq) = hrseg_split_f16x2(v);
    else st4(p.z + pix * p.ldz + 4 * L.cq, v);
  }
}

// MASK = some problem of the launch brings ReLU mask bytes.  Two instances because of REGISTERS: beside the 64-channel
// nine-tap weight gradient of the side stream (394 registers per lane of a SIMD) a wave of this kernel only fits under 112;
// the mask path costs 11 more (118) -- with one instance the UNet step, which has no masked layer at all, lost 2 ms of overlap.
template <bool MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_group_kernel(BnBwdG g) {
  __shared__ double red[256 * 8];
  int local, nblk;
  const hrseg_bn_bwd_t& p = g.p[bn_find(g.h, local, nblk)];
  const int nseg = p.nseg > 1 ? p.nseg : 1;
  // the |dy| slots the apply kernel raises with atomicMax start from zero: reset here, two launches earlier on the stream
  if (local == 0 && threadIdx.x < 64 && p.dy_absmax) p.dy_absmax[threadIdx.x] = 0.f;
  stats_body<MASK>(p.dz, p.lddz, p.z, p.ldz, p.relu, p.y, p.ldy, p.coef, p.npix, p.C, p.partial, local, p.nchunks, true, red,
                   nseg, p.relu_mask);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_group_kernel(BnBwdG g) {
  __shared__ double red[512];
  int local, nblk;
  const hrseg_bn_bwd_t& p = g.p[bn_find(g.h, local, nblk)];
  const int C = p.C, c = local * 16 + (threadIdx.x & 15);
  const int nseg = p.nseg > 1 ? p.nseg : 1, cps = p.nchunks / nseg;
  double* totals = p.partial + (size_t)p.nchunks * 2 * C;      // [nseg][2][C]
  double s_all = 0.0, sx_all = 0.0;
  for (int seg = 0; seg < nseg; ++seg) {
    double s, sx;
    if (seg) __syncthreads();                                   // red is reused
    reduce_chunks16(p.partial + (size_t)seg * cps * 2 * C, cps, C, c, s, sx, red);
    if (threadIdx.x < 16 && c < C) {
      if (p.sum_ranks > 1) {               // sums over all ranks -> this rank's share (means stay global: the apply divides by the local count)
        s /= (double)p.sum_ranks;
        sx /= (double)p.sum_ranks;
      }
      totals[(size_t)seg * 2 * C + c] = s;
      totals[(size_t)seg * 2 * C + C + c] = sx;
      s_all += s;
      sx_all += sx;
    }
  }
  if (threadIdx.x >= 16 || c >= C) return;
  if (p.dgamma) p.dgamma[c] += (float)sx_all;
  if (p.dbeta) p.dbeta[c] += (float)s_all;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_group_kernel(BnBwdG g, int eval_mode) {
  int local, nblk;
  const hrseg_bn_bwd_t& p = g.p[bn_find(g.h, local, nblk)];
  const int C = p.C;
  const Lanes L = make_lanes(C);
  float amax = 0.f;
  if (L.active) {                 // (no early return: every lane takes part in the max reduction below)
  const double* totals = p.partial + (size_t)p.nchunks * 2 * C;
  const f32x4 mean = ld4(p.coef + 4 * L.cq), rstd = ld4(p.coef + C + 4 * L.cq), scale = ld4(p.coef + 2 * C + 4 * L.cq);
  const f32x4 shift = ld4(p.coef + 3 * C + 4 * L.cq);
  const int nseg = p.nseg > 1 ? p.nseg : 1;
  const long seg_pix = p.npix / nseg;
  const float inv = eval_mode ? 0.f : (float)(1.0 / (double)seg_pix);
  for (int seg = 0; seg < nseg; ++seg) {
  f32x4 mg, mgx;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mg[j] = (float)(totals[(size_t)seg * 2 * C + 4 * L.cq + j]) * inv;
    mgx[j] = (float)(totals[(size_t)seg * 2 * C + C + 4 * L.cq + j]) * inv;
  }
  for (long pix = seg * seg_pix + (long)local * L.P + L.pl; pix < (seg + 1) * seg_pix; pix += (long)nblk * L.P) {
    f32x4 gg = ld4(p.dz + pix * p.lddz + 4 * L.cq);
    const f32x4 yv = ld4(p.y + pix * p.ldy + 4 * L.cq);
    if (p.relu && p.relu_mask) {
      const unsigned bm = p.relu_mask[pix * L.Q + L.cq];
#pragma unroll
      for (int j = 0; j < 4; ++j) gg[j] = ((bm >> j) & 1u) ? gg[j] : 0.f;
    } else if (p.relu) {
      const f32x4 zz = p.z ? ld4(p.z + pix * p.ldz + 4 * L.cq) : bn_affine(yv, scale, shift);
#pragma unroll
      for (int j = 0; j < 4; ++j) gg[j] = zz[j] > 0.f ? gg[j] : 0.f;
    }
    const f32x4 xh = (yv - mean) * rstd;
    const f32x4 dyv = scale * (gg - mg - xh * mgx);
    st4(p.dy + pix * p.lddy + 4 * L.cq, dyv);
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(dyv[j]));
    if (p.dres) {
      float* d = p.dres + pix * p.lddres + 4 * L.cq;
      st4(d, p.dres_accumulate ? ld4(d) + gg : gg);
    }
  }
  }
  }
  if (p.dy_absmax) {
    // max|dy| of the tensor into slot (block % 64) of a 64-entry array (non-negative floats order like their bit
    // patterns): one atomic per block, at most grid/64 per address -- a single address serialises ~4000 blocks
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(p.dy_absmax) + (blockIdx.x & 63), __float_as_uint(m));
    }
  }
}

// =========================================================================== C ABI
static long chunk_size(long npix, int nchunks) { return (npix + nchunks - 1) / nchunks; }

extern "C" int hrseg_bn_stats(const float* y, int ldy, long npix, int C, double* partial, int nchunks,
                              hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_bn_stats")) return e;
  HRSEG_CHECK_ARG(y && partial && npix > 0 && nchunks > 0 && ldy >= C, "hrseg_bn_stats: bad arguments");
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, y, ldy, npix, C, partial,
                     chunk_size(npix, nchunks));
  HRSEG_LAUNCH_CHECK("bn_stats");
  return 0;
}

extern "C" int hrseg_bn_finalize(const double* partial, int nchunks, long npix, int C, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float momentum, float eps, float* coef,
                                 hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(partial && coef && C > 0 && nchunks > 0 && npix > 0, "hrseg_bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, nchunks,
                     npix, C, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps,
                     coef);
  HRSEG_LAUNCH_CHECK("bn_finalize");
  return 0;
}

extern "C" int hrseg_bn_fold(const float* w, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, float eps, int Cout, int row, float* w_out, float* b_out,
                             hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(w && running_mean && running_var && w_out && b_out && Cout > 0 && row > 0, "hrseg_bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, bias, gamma, beta, running_mean,
                     running_var, eps, row, w_out, b_out);
  HRSEG_LAUNCH_CHECK("bn_fold");
  return 0;
}

extern "C" int hrseg_bn_eval_coef(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* coef, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(running_mean && running_var && coef && C > 0, "hrseg_bn_eval_coef: bad arguments");
  hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, coef);
  HRSEG_LAUNCH_CHECK("bn_eval_coef");
  return 0;
}

extern "C" int hrseg_bn_apply(const float* y, int ldy, const float* coef, const float* residual, int ldr, int relu,
                              float* z, int ldz, long npix, int C, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_bn_apply")) return e;
  HRSEG_CHECK_ARG(y && coef && z && npix > 0 && ldy >= C && ldz >= C && (ldy % 4 == 0) && (ldz % 4 == 0),
                  "hrseg_bn_apply: bad arguments");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, y, ldy, coef,
                     residual, ldr, relu, z, ldz, npix, C);
  HRSEG_LAUNCH_CHECK("bn_apply");
  return 0;
}

extern "C" int hrseg_bn_bwd_reduce(const float* dz, int lddz, const float* z, int ldz, int relu, const float* y,
                                   int ldy, const float* coef, long npix, int C, double* partial, int nchunks,
                                   hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_bn_bwd_reduce")) return e;
  HRSEG_CHECK_ARG(dz && y && coef && partial && (!relu || z) && npix > 0 && nchunks > 0,
                  "hrseg_bn_bwd_reduce: bad arguments");
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, dz, lddz, z, ldz, relu, y,
                     ldy, coef, npix, C, partial, chunk_size(npix, nchunks));
  HRSEG_LAUNCH_CHECK("bn_bwd_reduce");
  return 0;
}

extern "C" int hrseg_bn_bwd_apply(const double* partial, int nchunks, const float* dz, int lddz, const float* z,
                                  int ldz, int relu, const float* y, int ldy, const float* coef, const float* gamma,
                                  float* dgamma, float* dbeta, float* dy, int lddy, float* dres, int lddres,
                                  int dres_accumulate, long npix, int C, int eval_mode, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_bn_bwd_apply")) return e;
  (void)gamma;
  HRSEG_CHECK_ARG(partial && dz && y && coef && dy && (!relu || z) && npix > 0 && nchunks > 0,
                  "hrseg_bn_bwd_apply: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // totals live behind the nchunks partial slabs: the caller sizes `partial` as (nchunks+1)*2*C
  double* totals = (double*)partial + (size_t)nchunks * 2 * C;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, st, partial, totals, nchunks, C,
                     dgamma, dbeta);
  HRSEG_LAUNCH_CHECK("bn_bwd_finalize");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, st, totals, dz, lddz, z, ldz, relu,
                     y, ldy, coef, dy, lddy, dres, lddres, dres_accumulate, npix, C, eval_mode);
  HRSEG_LAUNCH_CHECK("bn_bwd_apply");
  return 0;
}

extern "C" int hrseg_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, int B, int Hi, int Wi, int C,
                                  hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_maxpool2_fwd")) return e;
  HRSEG_CHECK_ARG(x && y && B > 0 && Hi >= 2 && Wi >= 2, "hrseg_maxpool2_fwd: bad arguments");
  const long npix = (long)B * (Hi / 2) * (Wi / 2);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                     B, Hi, Wi, C);
  HRSEG_LAUNCH_CHECK("maxpool2_fwd");
  return 0;
}

extern "C" int hrseg_maxpool2_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx,
                                  int accumulate, int B, int Hi, int Wi, int C, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_maxpool2_bwd")) return e;
  HRSEG_CHECK_ARG(x && dy && dx && B > 0 && Hi >= 2 && Wi >= 2, "hrseg_maxpool2_bwd: bad arguments");
  const long ncell = (long)B * ((Hi + 1) / 2) * ((Wi + 1) / 2);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(elem_grid(ncell, C)), dim3(256), 0, (hipStream_t)stream, x, ldx, dy,
                     lddy, dx, lddx, accumulate, B, Hi, Wi, C);
  HRSEG_LAUNCH_CHECK("maxpool2_bwd");
  return 0;
}

extern "C" int hrseg_bilinear_fwd(const float* in, int ldin, int B, int Hi, int Wi, int C, float* out, int ldout,
                                  int Hout, int Wout, int Hr, int Wr, int py, int px, int align_corners,
                                  int accumulate, int relu, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_bilinear_fwd")) return e;
  HRSEG_CHECK_ARG(in && out && B > 0 && Hi > 0 && Wi > 0 && Hr > 0 && Wr > 0 && py >= 0 && px >= 0 &&
                      py + Hr <= Hout && px + Wr <= Wout,
                  "hrseg_bilinear_fwd: placed image %dx%d at (%d,%d) does not fit %dx%d", Hr, Wr, py, px, Hout, Wout);
  const long npix = (long)B * Hout * Wout;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, in, ldin, B, Hi,
                     Wi, C, out, ldout, Hout, Wout, Hr, Wr, py, px, resize_scale(Hi, Hr, align_corners),
                     resize_scale(Wi, Wr, align_corners), align_corners, accumulate, relu);
  HRSEG_LAUNCH_CHECK("bilinear_fwd");
  return 0;
}

extern "C" int hrseg_bilinear_bwd(const float* dout, int lddout, int B, int Hi, int Wi, int C, float* din, int lddin,
                                  int Hout, int Wout, int Hr, int Wr, int py, int px, int align_corners,
                                  int accumulate, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_bilinear_bwd")) return e;
  HRSEG_CHECK_ARG(dout && din && B > 0 && Hi > 0 && Wi > 0 && Hr > 0 && Wr > 0 && py >= 0 && px >= 0 &&
                      py + Hr <= Hout && px + Wr <= Wout,
                  "hrseg_bilinear_bwd: bad geometry");
  const long npix = (long)B * Hi * Wi;
  // row splits: enough blocks to fill the chip, at most half the footprint rows, Q*RS <= 256
  const int Q = C / 4;
  const int foot = (Hi > 1 && Hr > 1) ? (int)(2.0 * (Hr - 1) / (Hi - 1)) + 3 : Hr;
  int rs = 1;
  while (rs < 8 && Q * rs * 2 <= 256 && rs * 2 <= foot / 2 && npix / max(1, 256 / (Q * rs)) < 1024) rs *= 2;
  const int P2 = max(1, 256 / (Q * rs));
  long blocks = (npix + P2 - 1) / P2;
  if (blocks > 4096) blocks = 4096;
  const float sh = resize_scale(Hi, Hr, align_corners), sw = resize_scale(Wi, Wr, align_corners);
#define HRSEG_BIL(RS_) hipLaunchKernelGGL(bilinear_bwd_kernel<RS_>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, \
                                          dout, lddout, B, Hi, Wi, C, din, lddin, Hout, Wout, Hr, Wr, py, px, sh, sw,         \
                                          align_corners, accumulate)
  if (rs == 1) HRSEG_BIL(1); else if (rs == 2) HRSEG_BIL(2); else if (rs == 4) HRSEG_BIL(4); else HRSEG_BIL(8);
#undef HRSEG_BIL
  HRSEG_LAUNCH_CHECK("bilinear_bwd");
  return 0;
}

extern "C" int hrseg_fuse_sum(int n_same, const float* const* same, const int* ld_same, int n_low, const float* const* low,
                              const int* ld_low, const int* Hi, const int* Wi, float* out, int ldo, int B, int H, int W, int C,
                              int align_corners, int relu, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_fuse_sum")) return e;
  HRSEG_CHECK_ARG(n_same >= 1 && n_same <= 4 && n_low >= 0 && n_low <= 3 && same && ld_same && out && B > 0 && H > 0 && W > 0,
                  "hrseg_fuse_sum: 1..4 same-resolution terms and 0..3 low-resolution terms");
  HRSEG_CHECK_ARG(n_low == 0 || (low && ld_low && Hi && Wi), "hrseg_fuse_sum: low-resolution terms need their geometry");
  FuseSumArgs a{};
  a.n_same = n_same; a.n_low = n_low;
  for (int i = 0; i < n_same; ++i) {
    HRSEG_CHECK_ARG(same[i] && ld_same[i] >= C && ld_same[i] % 4 == 0, "hrseg_fuse_sum: bad same-resolution term %d", i);
    a.same[i] = same[i]; a.ld_same[i] = ld_same[i];
  }
  for (int j = 0; j < n_low; ++j) {
    HRSEG_CHECK_ARG(low[j] && ld_low[j] >= C && ld_low[j] % 4 == 0 && Hi[j] > 0 && Wi[j] > 0, "hrseg_fuse_sum: bad low-resolution term %d", j);
    a.low[j] = low[j]; a.ld_low[j] = ld_low[j]; a.Hi[j] = Hi[j]; a.Wi[j] = Wi[j];
    a.sh[j] = resize_scale(Hi[j], H, align_corners);
    a.sw[j] = resize_scale(Wi[j], W, align_corners);
  }
  a.out = out; a.ldo = ldo; a.B = B; a.H = H; a.W = W; a.C = C; a.align = align_corners; a.relu = relu;
  const long npix = (long)B * H * W;
  hipLaunchKernelGGL(fuse_sum_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, a);
  HRSEG_LAUNCH_CHECK("fuse_sum");
  return 0;
}

extern "C" int hrseg_add(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int relu, long npix,
                         int C, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_add")) return e;
  HRSEG_CHECK_ARG(a && b && out && npix > 0, "hrseg_add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo,
                     relu, npix, C);
  HRSEG_LAUNCH_CHECK("add");
  return 0;
}

extern "C" int hrseg_copy(const float* in, int ldin, float* out, int ldout, int accumulate, long npix, int C,
                          hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_copy")) return e;
  HRSEG_CHECK_ARG(in && out && npix > 0, "hrseg_copy: bad arguments");
  hipLaunchKernelGGL(copy_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, in, ldin, out, ldout,
                     accumulate, npix, C);
  HRSEG_LAUNCH_CHECK("copy");
  return 0;
}

extern "C" int hrseg_relu_bwd(const float* dz, int lddz, const float* z, int ldz, float* dx, int lddx, long npix,
                              int C, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_relu_bwd")) return e;
  HRSEG_CHECK_ARG(dz && z && dx && npix > 0, "hrseg_relu_bwd: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, dz, lddz, z, ldz,
                     dx, lddx, npix, C);
  HRSEG_LAUNCH_CHECK("relu_bwd");
  return 0;
}

extern "C" int hrseg_absmax(const float* x, int ldx, long npix, int C, float* out64, hrseg_stream_t stream) {
  if (int e = check_c(C, "hrseg_absmax")) return e;
  HRSEG_CHECK_ARG(x && out64 && npix > 0 && ldx >= C, "hrseg_absmax: bad arguments");
  hipLaunchKernelGGL(absmax_kernel, dim3(elem_grid(npix, C)), dim3(256), 0, (hipStream_t)stream, x, ldx, npix, C, out64);
  HRSEG_LAUNCH_CHECK("absmax");
  return 0;
}

extern "C" int hrseg_nchw_to_nhwc(const float* in, float* out, int ldout, int B, int C, int H, int W,
                                  hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(in && out && B > 0 && C > 0 && ldout >= C, "hrseg_nchw_to_nhwc: bad arguments");
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, ldout, B,
                     C, (long)H * W);
  HRSEG_LAUNCH_CHECK("nchw_to_nhwc");
  return 0;
}

extern "C" int hrseg_nhwc_to_nchw(const float* in, int ldin, float* out, int B, int C, int H, int W,
                                  hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(in && out && B > 0 && C > 0 && ldin >= C, "hrseg_nhwc_to_nchw: bad arguments");
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, in, ldin, out, B,
                     C, (long)H * W);
  HRSEG_LAUNCH_CHECK("nhwc_to_nchw");
  return 0;
}

extern "C" int hrseg_fill(float* p, float v, long n, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(p && n >= 0, "hrseg_fill: bad arguments");
  if (n == 0) return 0;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, v, n);
  HRSEG_LAUNCH_CHECK("fill");
  return 0;
}

extern "C" int hrseg_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, float bc1, float bc2, float gscale,
                           hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(p && g && m && v && n > 0, "hrseg_adamw: bad arguments");
  HRSEG_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                      ((uintptr_t)v % 16 == 0),
                  "hrseg_adamw: buffers must be 16-byte aligned");
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, 1.0f / sqrtf(bc2), gscale);
  HRSEG_LAUNCH_CHECK("adamw");
  return 0;
}

extern "C" int hrseg_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, float* state,
                               hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(p && g && m && v && hyper && state && n > 0, "hrseg_adamw_dev: bad arguments");
  HRSEG_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                      ((uintptr_t)v % 16 == 0),
                  "hrseg_adamw_dev: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, state, hyper);
  HRSEG_LAUNCH_CHECK("adam_tick");
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_dev_kernel, dim3((int)blocks), dim3(256), 0, st, p, g, m, v, n4, n, hyper, state);
  HRSEG_LAUNCH_CHECK("adamw_dev");
  return 0;
}

static int check_bn_fwd(const hrseg_bn_fwd_t& p, int training) {
  if (int e = check_c(p.C, "hrseg_bn_fwd_group")) return e;
  HRSEG_CHECK_ARG(p.y && p.z && p.coef && p.npix > 0 && p.ldy >= p.C && p.ldz >= p.C && p.ldy % 4 == 0 && p.ldz % 4 == 0,
                  "hrseg_bn_fwd_group: bad tensor arguments");
  HRSEG_CHECK_ARG(!training || (p.partial && p.nchunks > 0), "hrseg_bn_fwd_group: training needs partial/nchunks");
  HRSEG_CHECK_ARG(training || (p.running_mean && p.running_var), "hrseg_bn_fwd_group: eval needs running stats");
  return 0;
}

extern "C" int hrseg_bn_fwd_group(int n, const hrseg_bn_fwd_t* probs, int training, hrseg_stream_t stream) {
  return hrseg_bn_fwd_group_phases(n, probs, training, 7, stream);
}

extern "C" int hrseg_bn_fwd_group_phases(int n, const hrseg_bn_fwd_t* probs, int training, int phases, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(n >= 1 && n <= BN_MAXG && probs, "hrseg_bn_fwd_group: n must be 1..%d", BN_MAXG);
  HRSEG_CHECK_ARG(phases > 0 && phases <= 7, "hrseg_bn_fwd_group_phases: phases is a mask of bits 0..2");
  hipStream_t st = (hipStream_t)stream;
  BnFwdG g;
  g.h.n = n;
  for (int i = 0; i < n; ++i) {
    if (int e = check_bn_fwd(probs[i], training)) return e;
    g.p[i] = probs[i];
  }
  int end = 0;
  if (training) {
    if (phases & 1) {
      for (int i = 0; i < n; ++i) { end += probs[i].nchunks; g.h.blk_end[i] = end; }
      hipLaunchKernelGGL(bn_stats_group_kernel, dim3(end), dim3(256), 0, st, g);
      HRSEG_LAUNCH_CHECK("bn_stats_group");
    }
    if (phases & 2) {
      end = 0;
      for (int i = 0; i < n; ++i) { end += ceil_div(probs[i].C, 16); g.h.blk_end[i] = end; }
      hipLaunchKernelGGL(bn_finalize_group_kernel, dim3(end), dim3(256), 0, st, g);
      HRSEG_LAUNCH_CHECK("bn_finalize_group");
    }
  } else if (phases & 2) {
    for (int i = 0; i < n; ++i) { end += ceil_div(probs[i].C, 64); g.h.blk_end[i] = end; }
    hipLaunchKernelGGL(bn_eval_coef_group_kernel, dim3(end), dim3(64), 0, st, g);
    HRSEG_LAUNCH_CHECK("bn_eval_coef_group");
  }
  if (phases & 4) {
    end = 0;
    for (int i = 0; i < n; ++i) { end += elem_grid(probs[i].npix, probs[i].C); g.h.blk_end[i] = end; }
    hipLaunchKernelGGL(bn_apply_group_kernel, dim3(end), dim3(256), 0, st, g);
    HRSEG_LAUNCH_CHECK("bn_apply_group");
  }
  return 0;
}

extern "C" int hrseg_bn_bwd_group(int n, const hrseg_bn_bwd_t* probs, int eval_mode, hrseg_stream_t stream) {
  return hrseg_bn_bwd_group_phases(n, probs, eval_mode, 7, stream);
}

extern "C" int hrseg_bn_bwd_group_phases(int n, const hrseg_bn_bwd_t* probs, int eval_mode, int phases, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(n >= 1 && n <= BN_MAXG && probs, "hrseg_bn_bwd_group: n must be 1..%d", BN_MAXG);
  HRSEG_CHECK_ARG(phases > 0 && phases <= 7, "hrseg_bn_bwd_group_phases: phases is a mask of bits 0..2");
  hipStream_t st = (hipStream_t)stream;
  BnBwdG g;
  g.h.n = n;
  for (int i = 0; i < n; ++i) {
    const hrseg_bn_bwd_t& p = probs[i];
    if (int e = check_c(p.C, "hrseg_bn_bwd_group")) return e;
    HRSEG_CHECK_ARG(p.dz && p.y && p.coef && p.dy && p.partial && p.npix > 0 && p.nchunks > 0,
                    "hrseg_bn_bwd_group: bad arguments");
    HRSEG_CHECK_ARG(p.nseg <= 1 || (p.nchunks % p.nseg == 0 && p.npix % p.nseg == 0),
                    "hrseg_bn_bwd_group: nseg=%d must divide nchunks=%d and npix", p.nseg, p.nchunks);
    g.p[i] = p;
  }
  int end = 0;
  bool masked = false;
  for (int i = 0; i < n; ++i) masked = masked || (probs[i].relu && probs[i].relu_mask);
  if (phases & 1) {
    for (int i = 0; i < n; ++i) { end += probs[i].nchunks; g.h.blk_end[i] = end; }
    if (masked) hipLaunchKernelGGL(bn_bwd_reduce_group_kernel<true>, dim3(end), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(bn_bwd_reduce_group_kernel<false>, dim3(end), dim3(256), 0, st, g);
    HRSEG_LAUNCH_CHECK("bn_bwd_reduce_group");
  }
  if (phases & 2) {
    end = 0;
    for (int i = 0; i < n; ++i) { end += ceil_div(probs[i].C, 16); g.h.blk_end[i] = end; }
    hipLaunchKernelGGL(bn_bwd_finalize_group_kernel, dim3(end), dim3(256), 0, st, g);
    HRSEG_LAUNCH_CHECK("bn_bwd_finalize_group");
  }
  if (phases & 4) {
    end = 0;
    for (int i = 0; i < n; ++i) { end += elem_grid(probs[i].npix, probs[i].C); g.h.blk_end[i] = end; }
    hipLaunchKernelGGL(bn_bwd_apply_group_kernel, dim3(end), dim3(256), 0, st, g, eval_mode);
    HRSEG_LAUNCH_CHECK("bn_bwd_apply_group");
  }
  return 0;
}
