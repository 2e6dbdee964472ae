// FiLM + 1x1 heads, probability composition over the class tree, the fused
// CE + soft-Dice loss (forward partials, finalize, gradient), the consistency
// term and the per-step prediction / confusion-count kernel.  gfx950.
//
// Logits / probabilities / targets here are NCHW fp32 planes ([B,C,hw]) as the
// reference returns them; C per level is small (<= 16), so a thread owns one
// pixel and walks the C planes (each plane access is coalesced across lanes).
#include "common.h"

#define MAXC 16

struct Groups {
  int n;
  int parent[MAXC];
  int size[MAXC];
};

static int fill_groups(Groups& g, int ngroups, const int* parent, const int* size, int C, int Cprev, const char* who) {
  HRSEG_CHECK_ARG(ngroups >= 0 && ngroups <= MAXC && (ngroups == 0 || (parent && size)), "%s: bad group table", who);
  g.n = ngroups;
  int tot = 0;
  for (int i = 0; i < ngroups; ++i) {
    HRSEG_CHECK_ARG(parent[i] >= 0 && parent[i] < Cprev && size[i] > 0, "%s: group %d out of range", who, i);
    g.parent[i] = parent[i];
    g.size[i] = size[i];
    tot += size[i];
  }
  HRSEG_CHECK_ARG(tot == C || ngroups == 0, "%s: group sizes sum to %d, level has %d channels", who, tot, C);
  return 0;
}

// --------------------------------------------------------------------------- GAP over NCHW planes
__global__ __launch_bounds__(256) void gap_partial_kernel(const float* __restrict__ p, double* __restrict__ part,
                                                          long hw, int nsplit) {
  __shared__ double red[4];
  const int row = blockIdx.y, sp = blockIdx.x;
  const long per = (hw + nsplit - 1) / nsplit;
  const long lo = sp * per, hi = (lo + per < hw) ? lo + per : hw;
  const float* r = p + (size_t)row * hw;
  float s = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) s += r[i];
  double d = wave_sum_d((double)s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)row * nsplit + sp] = red[0] + red[1] + red[2] + red[3];
}
__global__ void gap_final_kernel(const double* __restrict__ part, float* __restrict__ cond, int rows, int nsplit,
                                 long hw) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  double s = 0.0;
  for (int i = 0; i < nsplit; ++i) s += part[(size_t)r * nsplit + i];
  cond[r] = (float)(s / (double)hw);
}

// --------------------------------------------------------------------------- FiLM linear (tiny)
__global__ void film_linear_fwd_kernel(const float* cond, const float* wl, const float* bl, float* gb, int B, int Cc,
                                       int F2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * F2) return;
  const int b = i / F2, j = i - b * F2;
  float s = bl[j];
  for (int c = 0; c < Cc; ++c) s = fmaf(cond[b * Cc + c], wl[j * Cc + c], s);
  gb[i] = s;
}
// dcond[b][c] = scale * sum_j dgb[b][j] wl[j][c];  dwl[j][c] += sum_b dgb[b][j] cond[b][c];  dbl[j] += sum_b dgb[b][j]
__global__ void film_linear_bwd_kernel(const float* cond, const float* wl, const float* dgb, float* dcond, float* dwl,
                                       float* dbl, int B, int Cc, int F2, float scale) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < F2) {
    float sb = 0.f;
    for (int b = 0; b < B; ++b) sb += dgb[b * F2 + j];
    if (dbl) dbl[j] += sb;
    if (dwl)
      for (int c = 0; c < Cc; ++c) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(dgb[b * F2 + j], cond[b * Cc + c], s);
        dwl[j * Cc + c] += s;
      }
  }
  if (dcond && blockIdx.x == 0) {
    // one wave per output element, lanes stride the 2F-long dot product (one thread per element walked it alone:
    // 1440 dependent steps, 180 us for a 32-element result)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int i = wave; i < B * Cc; i += nwave) {
      const int b = i / Cc, c = i - b * Cc;
      float s = 0.f;
      for (int jj = lane; jj < F2; jj += 64) s = fmaf(dgb[b * F2 + jj], wl[jj * Cc + c], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) dcond[i] = s * scale;
    }
  }
}

// --------------------------------------------------------------------------- head forward
// z[pix][c] = sum_k W[c][k] (f[pix][k] g[b][k] + be[b][k]) + bias[c].  Effective per-sample
// weights W*g and bias + W.be are built in LDS; LP lanes share one pixel row.
template <int LP, int CO>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ f, int ldf,
                                                       const float* __restrict__ gb, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ z, int ldz,
                                                       long hw, int F, int Cout) {
  extern __shared__ float sm[];  // weff[CO][F], beff[CO]
  float* weff = sm;
  float* beff = sm + CO * F;
  const int b = blockIdx.y;
  const float* gam = gb ? gb + (size_t)b * 2 * F : nullptr;
  for (int i = threadIdx.x; i < CO * F; i += 256) {
    const int c = i / F, k = i - c * F;
    weff[i] = (c < Cout) ? w[c * F + k] * (gam ? gam[k] : 1.f) : 0.f;
  }
  // beff[c] = bias[c] + sum_k w[c][k] * beta[k]: all 256 threads take part (four threads walking F elements each was
  // a 720-step serial chain at the head of every block: most of this kernel's time with FiLM on)
  {
    __shared__ float red[CO][4];
    float part[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      part[c] = 0.f;
      if (gam && c < Cout)
        for (int k = threadIdx.x; k < F; k += 256) part[c] = fmaf(w[c * F + k], gam[F + k], part[c]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part[c] += __shfl_xor(part[c], o, 64);
      if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = part[c];
    }
    __syncthreads();
    if (threadIdx.x < CO) {
      const int c = threadIdx.x;
      beff[c] = ((c < Cout && bias) ? bias[c] : 0.f) + ((red[c][0] + red[c][1]) + (red[c][2] + red[c][3]));
    }
  }
  __syncthreads();
  constexpr int PPB = 256 / LP;  // pixels per block iteration
  const int sub = threadIdx.x % LP, pg = threadIdx.x / LP;
  const int Q = F >> 2;
  if (Q <= 3 * LP) {
    // a lane reads the same (at most three) 4-channel groups of every pixel: its slice of the effective weights lives
    // in registers for the whole block, and the pixel loop is global loads + FMAs only (reading the weights from LDS
    // per pixel was four ds_read_b128 per 16 bytes of features: 217 us for the 553 MB of the HRNet feature map)
    f32x4 wr[3][CO];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const int q = sub + j * LP;
        wr[j][c] = (q < Q) ? *reinterpret_cast<const f32x4*>(weff + c * F + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    for (long pix = (long)blockIdx.x * PPB + pg; pix < hw; pix += (long)gridDim.x * PPB) {
      const float* row = f + ((size_t)b * hw + pix) * ldf;
      f32x4 v[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int q = sub + j * LP;
        v[j] = (q < Q) ? *reinterpret_cast<const f32x4*>(row + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      float acc[CO];
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        acc[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[c] += v[j][0] * wr[j][c][0] + v[j][1] * wr[j][c][1] + v[j][2] * wr[j][c][2] + v[j][3] * wr[j][c][3];
      }
#pragma unroll
      for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int o = LP / 2; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
      if (sub == 0) {
        float* o = z + ((size_t)b * hw + pix) * ldz;
#pragma unroll
        for (int c = 0; c < CO; ++c)
          if (c < Cout) o[c] = acc[c] + beff[c];
      }
    }
    return;
  }
  for (long pix = (long)blockIdx.x * PPB + pg; pix < hw; pix += (long)gridDim.x * PPB) {
    const float* row = f + ((size_t)b * hw + pix) * ldf;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    for (int q = sub; q < Q; q += LP) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q);
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(weff + c * F + 4 * q);
        acc[c] += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3];
      }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
      for (int o = LP / 2; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    if (sub == 0) {
      float* o = z + ((size_t)b * hw + pix) * ldz;
#pragma unroll
      for (int c = 0; c < CO; ++c)
        if (c < Cout) o[c] = acc[c] + beff[c];
    }
  }
}

// head backward: thread owns channel quad cq of pixel lane pl; block = (pixel chunk, sample)
template <int CO>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ f, int ldf,
                                                       const float* __restrict__ gb, const float* __restrict__ w,
                                                       const float* __restrict__ dz, int lddz,
                                                       float* __restrict__ df, int lddf, int df_acc,
                                                       float* __restrict__ dw, float* __restrict__ dbias,
                                                       float* __restrict__ dgb, long hw, int F, int Cout,
                                                       long pix_per_block, int b0) {
  __shared__ float red[256];
  const int Q = F >> 2;
  const int P = (Q >= 256) ? 1 : 256 / Q;
  const int cq = threadIdx.x % Q, pl = threadIdx.x / Q;
  const bool active = pl < P;
  const int b = blockIdx.y + b0;
  const long lo = (long)blockIdx.x * pix_per_block;
  const long hi = (lo + pix_per_block < hw) ? lo + pix_per_block : hw;
  f32x4 gam = {1.f, 1.f, 1.f, 1.f}, bet = {0.f, 0.f, 0.f, 0.f};
  f32x4 wq[CO];
  f32x4 a_dw[CO], a_dg = {0.f, 0.f, 0.f, 0.f}, a_db = {0.f, 0.f, 0.f, 0.f};
  float a_bias[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    wq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    a_dw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    a_bias[c] = 0.f;
  }
  if (active) {
    if (gb) {
      gam = *reinterpret_cast<const f32x4*>(gb + (size_t)b * 2 * F + 4 * cq);
      bet = *reinterpret_cast<const f32x4*>(gb + (size_t)b * 2 * F + F + 4 * cq);
    }
#pragma unroll
    for (int c = 0; c < CO; ++c)
      if (c < Cout) wq[c] = *reinterpret_cast<const f32x4*>(w + c * F + 4 * cq);
    // four pixels per iteration, every load of the four before the arithmetic: with one channel group per thread and
    // one pixel per iteration the loop had a single 16-byte load in flight per thread
    constexpr int U = 4;
    auto one = [&](const f32x4& fv, const float (&g)[CO], const f32x4& dold, size_t r) {
      const f32x4 fm = fv * gam + bet;
      f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        u += wq[c] * g[c];
        a_dw[c] += fm * g[c];
        if (cq == 0) a_bias[c] += g[c];
      }
      a_dg += fv * u;
      a_db += u;
      if (df) *reinterpret_cast<f32x4*>(df + r * lddf + 4 * cq) = dold + gam * u;
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool racc = df && df_acc;
    long pix = lo + pl;
    for (; pix + (long)(U - 1) * P < hi; pix += (long)U * P) {
      f32x4 fv[U], dold[U];
      float g[U][CO];
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const size_t r = (size_t)b * hw + pix + (long)i * P;
        fv[i] = *reinterpret_cast<const f32x4*>(f + r * ldf + 4 * cq);
        dold[i] = racc ? *reinterpret_cast<const f32x4*>(df + r * lddf + 4 * cq) : zero;
#pragma unroll
        for (int c = 0; c < CO; ++c) g[i][c] = (c < Cout) ? dz[r * lddz + c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < U; ++i) one(fv[i], g[i], dold[i], (size_t)b * hw + pix + (long)i * P);
    }
    for (; pix < hi; pix += P) {
      const size_t r = (size_t)b * hw + pix;
      float g[CO];
#pragma unroll
      for (int c = 0; c < CO; ++c) g[c] = (c < Cout) ? dz[r * lddz + c] : 0.f;
      one(*reinterpret_cast<const f32x4*>(f + r * ldf + 4 * cq), g, racc ? *reinterpret_cast<const f32x4*>(df + r * lddf + 4 * cq) : zero, r);
    }
  }
  // reduce over the pixel lanes of the block, then one atomic per (channel, output)
  auto reduce_add = [&](float v, float* dst) {
    red[threadIdx.x] = v;
    __syncthreads();
    if (active && pl == 0) {
      float s = 0.f;
      for (int q = 0; q < P; ++q) s += red[q * Q + cq];
      atomicAdd(dst, s);
    }
    __syncthreads();
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int c = 0; c < CO; ++c)
      if (c < Cout) reduce_add(a_dw[c][j], dw + c * F + 4 * cq + j);
    if (dgb) {
      reduce_add(a_dg[j], dgb + (size_t)b * 2 * F + 4 * cq + j);
      reduce_add(a_db[j], dgb + (size_t)b * 2 * F + F + 4 * cq + j);
    }
  }
  if (dbias) {
#pragma unroll
    for (int c = 0; c < CO; ++c)
      if (c < Cout) {
        red[threadIdx.x] = (active && cq == 0) ? a_bias[c] : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
          float s = 0.f;
          for (int q = 0; q < P; ++q) s += red[q * Q];
          atomicAdd(dbias + c, s);
        }
        __syncthreads();
      }
  }
}

// --------------------------------------------------------------------------- logits resize NHWC(low) -> NCHW(full)
__device__ __forceinline__ void src_index_ac(int o, float scale, int in_size, int align, int& i0, int& i1, float& l0,
                                             float& l1) {
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
__global__ void logits_up_fwd_kernel(const float* __restrict__ in, int ldin, int B, int Hi, int Wi, int C,
                                     float* __restrict__ out, int Ho, int Wo, float sh, float sw, int align) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long hw = (long)Ho * Wo;
  if (i >= B * hw) return;
  const int b = (int)(i / hw);
  const long p = i - b * hw;
  const int oy = (int)(p / Wo), ox = (int)(p - (long)oy * Wo);
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  src_index_ac(oy, sh, Hi, align, y0, y1, ly0, ly1);
  src_index_ac(ox, sw, Wi, align, x0, x1, lx0, lx1);
  const float* base = in + (size_t)b * Hi * Wi * ldin;
  for (int c = 0; c < C; ++c) {
    const float v00 = base[((size_t)y0 * Wi + x0) * ldin + c], v01 = base[((size_t)y0 * Wi + x1) * ldin + c];
    const float v10 = base[((size_t)y1 * Wi + x0) * ldin + c], v11 = base[((size_t)y1 * Wi + x1) * ldin + c];
    out[((size_t)b * C + c) * hw + p] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
}
// din[b,iy,ix,c] = sum over output pixels touching (iy,ix); one thread per (b,iy,ix), all C channels
__global__ void logits_up_bwd_kernel(const float* __restrict__ dout, int B, int Hi, int Wi, int C,
                                     float* __restrict__ din, int lddin, int Ho, int Wo, float sh, float sw,
                                     int align) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Hi * Wi) return;
  const int b = (int)(i / ((long)Hi * Wi));
  const int rem = (int)(i - (long)b * Hi * Wi);
  const int iy = rem / Wi, ix = rem - iy * Wi;
  const long hw = (long)Ho * Wo;
  const float ish = sh > 0.f ? 1.f / sh : 0.f, isw = sw > 0.f ? 1.f / sw : 0.f;
  const float hf = align ? 0.f : 0.5f;
  int oy_lo = 0, oy_hi = Ho - 1, ox_lo = 0, ox_hi = Wo - 1;
  if (sh > 0.f) {
    oy_lo = max(0, (int)floorf(((float)iy - 1.f + hf) * ish - hf) - 1);
    oy_hi = min(Ho - 1, (int)ceilf(((float)iy + 1.f + hf) * ish - hf) + 1);
  }
  if (sw > 0.f) {
    ox_lo = max(0, (int)floorf(((float)ix - 1.f + hf) * isw - hf) - 1);
    ox_hi = min(Wo - 1, (int)ceilf(((float)ix + 1.f + hf) * isw - hf) + 1);
  }
  float s[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) s[c] = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    int y0, y1;
    float ly0, ly1;
    src_index_ac(oy, sh, Hi, align, y0, y1, ly0, ly1);
    const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      int x0, x1;
      float lx0, lx1;
      src_index_ac(ox, sw, Wi, align, x0, x1, lx0, lx1);
      const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
      if (wx == 0.f) continue;
      const float wgt = wy * wx;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) s[c] = fmaf(wgt, dout[((size_t)b * C + c) * hw + (size_t)oy * Wo + ox], s[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) din[i * lddin + c] = s[c];
}

// --------------------------------------------------------------------------- composition
__global__ void sigmoid_fwd_kernel(const float* __restrict__ z, float* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 1.f / (1.f + expf(-z[i]));
}
// dz (=|+=) dp * p (1-p); dp element (b,c,i) at b*sb + c*sc + i*si
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dp, long sb, long sc, long si,
                                   const float* __restrict__ z, float* __restrict__ dz, int acc, int C, long hw,
                                   long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long bc = i / hw, pix = i - bc * hw;
  const long b = bc / C, c = bc - b * C;
  const float p = 1.f / (1.f + expf(-z[i]));
  const float g = dp[b * sb + c * sc + pix * si] * p * (1.f - p);
  dz[i] = acc ? dz[i] + g : g;
}

#define EPS_GATE 1e-6f
__global__ void compose_fwd_kernel(const float* __restrict__ z, const float* __restrict__ pprev,
                                   float* __restrict__ p, int C, int Cprev, long hw, long n, Groups g) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long b = i / hw, pix = i - b * hw;
  const float* zb = z + (size_t)b * C * hw + pix;
  float* pb = p + (size_t)b * C * hw + pix;
  int start = 0;
  for (int gi = 0; gi < g.n; ++gi) {
    const int sz = g.size[gi];
    const float pp = pprev[((size_t)b * Cprev + g.parent[gi]) * hw + pix];
    const float lg = logf(pp + EPS_GATE);
    float m = -INFINITY;
    for (int c = 0; c < sz; ++c) m = fmaxf(m, zb[(size_t)(start + c) * hw] + lg);
    float s = 0.f;
    for (int c = 0; c < sz; ++c) s += expf(zb[(size_t)(start + c) * hw] + lg - m);
    const float inv = 1.f / s;
    for (int c = 0; c < sz; ++c) pb[(size_t)(start + c) * hw] = pp * (expf(zb[(size_t)(start + c) * hw] + lg - m) * inv);
    start += sz;
  }
}
__global__ void compose_bwd_kernel(const float* __restrict__ dp, long sb, long sc, long si,
                                   const float* __restrict__ z, const float* __restrict__ pprev,
                                   float* __restrict__ dz, int dz_acc, float* __restrict__ dpprev, int dpp_acc,
                                   int C, int Cprev, long hw, long n, Groups g) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long b = i / hw, pix = i - b * hw;
  const float* zb = z + (size_t)b * C * hw + pix;
  float dpar[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) dpar[c] = 0.f;
  int start = 0;
  for (int gi = 0; gi < g.n; ++gi) {
    const int sz = g.size[gi];
    const float pp = pprev[((size_t)b * Cprev + g.parent[gi]) * hw + pix];
    const float lg = logf(pp + EPS_GATE);
    float m = -INFINITY;
    for (int c = 0; c < sz; ++c) m = fmaxf(m, zb[(size_t)(start + c) * hw] + lg);
    float s = 0.f;
    for (int c = 0; c < sz; ++c) s += expf(zb[(size_t)(start + c) * hw] + lg - m);
    const float inv = 1.f / s;
    // dq_c = dP_c * pp ; ds_c = q_c (dq_c - sum_j q_j dq_j)
    float dot = 0.f, dpp = 0.f;
    for (int c = 0; c < sz; ++c) {
      const float q = expf(zb[(size_t)(start + c) * hw] + lg - m) * inv;
      const float d = dp[b * sb + (start + c) * sc + pix * si];
      dot += q * d * pp;
      dpp += d * q;
    }
    float sum_ds = 0.f;
    for (int c = 0; c < sz; ++c) {
      const float q = expf(zb[(size_t)(start + c) * hw] + lg - m) * inv;
      const float d = dp[b * sb + (start + c) * sc + pix * si];
      const float ds = q * (d * pp - dot);
      sum_ds += ds;
      if (dz) {
        float* o = dz + ((size_t)b * C + start + c) * hw + pix;
        *o = dz_acc ? *o + ds : ds;
      }
    }
    dpp += sum_ds / (pp + EPS_GATE);
    // several groups never share a parent, but keep += for safety
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c == g.parent[gi]) dpar[c] += dpp;
    start += sz;
  }
  if (dpprev) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < Cprev) {
        float* o = dpprev + ((size_t)b * Cprev + c) * hw + pix;
        *o = dpp_acc ? *o + dpar[c] : dpar[c];
      }
  }
}

__global__ void zero_f64_kernel(double* p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.0;
}

// --------------------------------------------------------------------------- loss
// per (b,c): {n, sum t logp, sum p t, sum p, sum t} over pixels with t != -1
template <int CT>
__global__ __launch_bounds__(256) void loss_partials_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                            double* __restrict__ partial, int C, long hw,
                                                            long pix_per_block) {
  __shared__ float red[4][CT * 5];
  const int b = blockIdx.y;
  const long lo = (long)blockIdx.x * pix_per_block;
  const long hi = (lo + pix_per_block < hw) ? lo + pix_per_block : hw;
  float acc[CT][5];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[c][k] = 0.f;
  const float* zb = z + (size_t)b * C * hw;
  const float* tb = t + (size_t)b * C * hw;
  for (long pix = lo + threadIdx.x; pix < hi; pix += 256) {
    float zz[CT], tt[CT];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      zz[c] = (c < C) ? zb[(size_t)c * hw + pix] : -INFINITY;
      tt[c] = (c < C) ? tb[(size_t)c * hw + pix] : -1.f;
      m = fmaxf(m, zz[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      zz[c] = zz[c] - m;
      s += (c < C) ? expf(zz[c]) : 0.f;
    }
    const float ls = logf(s), inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (c < C && tt[c] != -1.f) {
        const float logp = zz[c] - ls, p = expf(zz[c]) * inv;
        acc[c][0] += 1.f;
        acc[c][1] += tt[c] * logp;
        acc[c][2] += p * tt[c];
        acc[c][3] += p;
        acc[c][4] += tt[c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float v = wave_sum(acc[c][k]);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c * 5 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < C * 5) {
    const double v = (double)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(partial + (size_t)b * C * 5 + threadIdx.x, v);
  }
}

// out[0]=CE out[1]=Dice out[2]=#valid dice items ; coef[b][c] = {ce, dice_a, dice_b, 0}
__global__ void loss_finalize_kernel(const double* __restrict__ partial, const float* __restrict__ w, int B, int C,
                                     float* __restrict__ out, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double ce = 0.0, dice = 0.0;
  int nvalid = 0;
  for (int b = 0; b < B; ++b) {
    double U = 0.0;
    for (int c = 0; c < C; ++c) U += (double)w[c] * (partial[(b * C + c) * 5 + 3] + partial[(b * C + c) * 5 + 4]);
    if (U != 0.0) ++nvalid;
  }
  for (int b = 0; b < B; ++b) {
    bool ok = true;
    double item = 0.0, I = 0.0, U = 0.0;
    for (int c = 0; c < C; ++c) {
      const double* q = partial + (b * C + c) * 5;
      if (q[0] <= 0.0) ok = false;
      else item += -(double)w[c] * q[1] / q[0];
      I += (double)w[c] * q[2];
      U += (double)w[c] * (q[3] + q[4]);
    }
    ce += ok ? item / C : 1.0;
    const bool dv = (U != 0.0);
    if (dv) dice += 1.0 - 2.0 * I / U;
    for (int c = 0; c < C; ++c) {
      const double* q = partial + (b * C + c) * 5;
      float* k = coef + (b * C + c) * 4;
      k[0] = ok ? (float)(-(double)w[c] / (q[0] * C * B)) : 0.f;
      k[1] = dv ? (float)(-2.0 * w[c] / (U * nvalid)) : 0.f;
      k[2] = dv ? (float)(2.0 * I * w[c] / (U * U * nvalid)) : 0.f;
      k[3] = 0.f;
    }
  }
  out[0] = (float)(ce / B);
  out[1] = nvalid ? (float)(dice / nvalid) : 0.f;
  out[2] = (float)nvalid;
}

template <int CT>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                       const float* __restrict__ coef, const float* __restrict__ g,
                                                       float* __restrict__ dz, int acc, int C, long hw) {
  const int b = blockIdx.y;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= hw) return;
  const float gce = g[0], gdice = g[1];
  const float* zb = z + (size_t)b * C * hw + pix;
  const float* tb = t + (size_t)b * C * hw + pix;
  float zz[CT], tt[CT], p[CT];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    zz[c] = (c < C) ? zb[(size_t)c * hw] : -INFINITY;
    tt[c] = (c < C) ? tb[(size_t)c * hw] : -1.f;
    m = fmaxf(m, zz[c]);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    p[c] = (c < C) ? expf(zz[c] - m) : 0.f;
    s += p[c];
  }
  const float inv = 1.f / s;
  float dlogp[CT], dpr[CT], sum_dlogp = 0.f, dot = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    p[c] *= inv;
    dlogp[c] = 0.f;
    dpr[c] = 0.f;
    if (c < C && tt[c] != -1.f) {
      const float* k = coef + ((size_t)b * C + c) * 4;
      dlogp[c] = gce * k[0] * tt[c];
      dpr[c] = gdice * (k[1] * tt[c] + k[2]);
    }
    sum_dlogp += dlogp[c];
    dot += p[c] * dpr[c];
  }
  float* db = dz + (size_t)b * C * hw + pix;
#pragma unroll
  for (int c = 0; c < CT; ++c)
    if (c < C) {
      const float v = dlogp[c] - p[c] * sum_dlogp + p[c] * (dpr[c] - dot);
      db[(size_t)c * hw] = acc ? db[(size_t)c * hw] + v : v;
    }
}

// sum over pixels of | sum_{c in group} P[b,c] - Pprev[b,parent] | per group
__global__ __launch_bounds__(256) void consistency_kernel(const float* __restrict__ p, const float* __restrict__ pprev,
                                                          double* __restrict__ out, int C, int Cprev, long hw, long n,
                                                          Groups g) {
  __shared__ float red[4][MAXC];
  float acc[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) acc[k] = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / hw, pix = i - b * hw;
    int start = 0;
#pragma unroll
    for (int gi = 0; gi < MAXC; ++gi)
      if (gi < g.n) {
        float s = 0.f;
        for (int c = 0; c < g.size[gi]; ++c) s += p[((size_t)b * C + start + c) * hw + pix];
        acc[gi] += fabsf(s - pprev[((size_t)b * Cprev + g.parent[gi]) * hw + pix]);
        start += g.size[gi];
      }
  }
#pragma unroll
  for (int gi = 0; gi < MAXC; ++gi) {
    const float v = wave_sum(acc[gi]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][gi] = v;
  }
  __syncthreads();
  if (threadIdx.x < g.n)
    atomicAdd(out + threadIdx.x,
              (double)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// gradient of sum_g (scale_g) * sum_pix |sum_{c in g} P[c] - Pprev[parent_g]|:
// dP[c] = g*scale*sign(diff), dPprev[parent] -= g*scale*sign(diff); upstream scalar g on the device
__global__ void consistency_bwd_kernel(const float* __restrict__ p, const float* __restrict__ pprev,
                                       const float* __restrict__ gup, float scale, float* __restrict__ dp,
                                       float* __restrict__ dpprev, int C, int Cprev, long hw, long n, Groups g) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long b = i / hw, pix = i - b * hw;
  const float gs = gup[0] * scale;
  float dpar[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) dpar[c] = 0.f;
  int start = 0;
  for (int gi = 0; gi < g.n; ++gi) {
    float s = 0.f;
    for (int c = 0; c < g.size[gi]; ++c) s += p[((size_t)b * C + start + c) * hw + pix];
    const float d = s - pprev[((size_t)b * Cprev + g.parent[gi]) * hw + pix];
    const float sg = (d > 0.f) ? gs : (d < 0.f ? -gs : 0.f);
    for (int c = 0; c < g.size[gi]; ++c) dp[((size_t)b * C + start + c) * hw + pix] = sg;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c == g.parent[gi]) dpar[c] -= sg;
    start += g.size[gi];
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < Cprev) dpprev[((size_t)b * Cprev + c) * hw + pix] = dpar[c];
}

// --------------------------------------------------------------------------- grouped conditional KL (opt-in stabiliser)
// Restates the commented-out `grouped_conditional_kl` of the reference (Metrics/losses.py:180-210): per parent group g
// of the level, Q = softmax_c(z_c + log(P_parent + 1e-6)).clamp_min(1e-8) over the group's children and
// KL(Q || Uniform) = Q * (log Q - log(1/g)), `.mean()` over [B, g, H, W]; the level's value is the mean over its groups.
// (The log-bias is the same for every child of a group, so Q = softmax(z) and P_parent receives no gradient.)
// out[gi] += sum over pixels and children of Qc * (log Qc + log g), double.
__global__ __launch_bounds__(256) void group_kl_kernel(const float* __restrict__ z, const float* __restrict__ pprev,
                                                       double* __restrict__ out, int C, int Cprev, long hw, long n, Groups g) {
  __shared__ float red[4][MAXC];
  float acc[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) acc[k] = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / hw, pix = i - b * hw;
    int start = 0;
#pragma unroll
    for (int gi = 0; gi < MAXC; ++gi)
      if (gi < g.n) {
        const int gs = g.size[gi];
        const float lb = logf(pprev[((size_t)b * Cprev + g.parent[gi]) * hw + pix] + 1e-6f);
        float v[MAXC], m = -INFINITY;
        for (int c = 0; c < gs; ++c) {
          v[c] = z[((size_t)b * C + start + c) * hw + pix] + lb;
          m = fmaxf(m, v[c]);
        }
        float den = 0.f;
        for (int c = 0; c < gs; ++c) { v[c] = expf(v[c] - m); den += v[c]; }
        const float lg = logf((float)gs);
        float kl = 0.f;
        for (int c = 0; c < gs; ++c) {
          const float q = fmaxf(v[c] / den, 1e-8f);
          kl += q * (logf(q) + lg);
        }
        acc[gi] += kl;
        start += gs;
      }
  }
#pragma unroll
  for (int gi = 0; gi < MAXC; ++gi) {
    const float v = wave_sum(acc[gi]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][gi] = v;
  }
  __syncthreads();
  if (threadIdx.x < g.n)
    atomicAdd(out + threadIdx.x,
              (double)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dz of  gup * sum_g scale_g * sum_{pix,c} Qc (log Qc + log g),  scale_g = scale / (size_g)  (the .mean() over the
// channel dimension; `scale` carries 1 / (B*H*W*ngroups)).  With a_c = [Q_c >= 1e-8] (log Qc_c + 1 + log g):
// dz_j = Q_j (a_j - sum_c a_c Q_c).
__global__ void group_kl_bwd_kernel(const float* __restrict__ z, const float* __restrict__ pprev,
                                    const float* __restrict__ gup, float scale, float* __restrict__ dz, int C, int Cprev,
                                    long hw, long n, Groups g) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long b = i / hw, pix = i - b * hw;
  const float gsc = gup[0] * scale;
  int start = 0;
  for (int gi = 0; gi < g.n; ++gi) {
    const int gs = g.size[gi];
    const float lb = logf(pprev[((size_t)b * Cprev + g.parent[gi]) * hw + pix] + 1e-6f);
    float v[MAXC], m = -INFINITY;
    for (int c = 0; c < gs; ++c) {
      v[c] = z[((size_t)b * C + start + c) * hw + pix] + lb;
      m = fmaxf(m, v[c]);
    }
    float den = 0.f;
    for (int c = 0; c < gs; ++c) { v[c] = expf(v[c] - m); den += v[c]; }
    const float lg = logf((float)gs);
    float a[MAXC], dot = 0.f;
    for (int c = 0; c < gs; ++c) {
      const float q = v[c] / den;
      v[c] = q;
      a[c] = (q >= 1e-8f) ? (logf(q) + 1.f + lg) : 0.f;
      dot += a[c] * q;
    }
    const float sc = gsc / (float)gs;
    for (int c = 0; c < gs; ++c) dz[((size_t)b * C + start + c) * hw + pix] = sc * v[c] * (a[c] - dot);
    start += gs;
  }
}

// argmax one-hot (masked by t != -1) and confusion counts cm[target][pred]
template <int CT>
__global__ __launch_bounds__(256) void predict_metrics_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                              float* __restrict__ onehot,
                                                              unsigned long long* __restrict__ cm, int C, long hw,
                                                              long n, int child, int mask_pred) {
  __shared__ unsigned int hist[(MAXC + 1) * (MAXC + 1)];
  const int K = C + (child ? 1 : 0);
  for (int i = threadIdx.x; i < K * K; i += 256) hist[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / hw, pix = i - b * hw;
    const float* zb = z + (size_t)b * C * hw + pix;
    const float* tb = t + (size_t)b * C * hw + pix;
    int am = 0;
    float best = zb[0], zsum = 0.f;
    float tt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      tt[c] = (c < C) ? tb[(size_t)c * hw] : 0.f;
      if (c < C) {
        const float v = zb[(size_t)c * hw];
        zsum += v;
        if (c > 0 && v > best) { best = v; am = c; }
      }
    }
    // prediction label
    int pred;
    if (mask_pred) {
      bool kept = false;  // the arg-max channel survives unless its own target is -1
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c == am) kept = (tt[c] != -1.f);
      if (onehot) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
          if (c < C) onehot[((size_t)b * C + c) * hw + pix] = (c == am && tt[c] != -1.f) ? 1.f : 0.f;
      }
      pred = child ? (kept ? am + 1 : 0) : (kept ? am : 0);
    } else {
      pred = child ? ((zsum == 0.f) ? 0 : am + 1) : am;
    }
    // target label: arg-max (first) of the target with -1 -> 0 when masking, raw otherwise
    int tl = 0;
    float tbest = -INFINITY, tsum = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
        const float v = (mask_pred && tt[c] == -1.f) ? 0.f : tt[c];
        tsum += v;
        if (v > tbest) { tbest = v; tl = c; }
      }
    if (child) {
      // background channel (sum == 0) is prepended: it wins the arg-max when it is 1 (ties -> first)
      const float bg = (tsum == 0.f) ? 1.f : 0.f;
      tl = (bg >= tbest) ? 0 : tl + 1;
    }
    atomicAdd(&hist[tl * K + pred], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += 256)
    if (hist[i]) atomicAdd(cm + i, (unsigned long long)hist[i]);
}

// =========================================================================== C ABI
extern "C" int hrseg_gap_nchw(const float* p, float* cond, double* scratch, int BC, long hw, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(p && cond && scratch && BC > 0 && hw > 0, "hrseg_gap_nchw: bad arguments");
  const int nsplit = 64;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gap_partial_kernel, dim3(nsplit, BC), dim3(256), 0, st, p, scratch, hw, nsplit);
  HRSEG_LAUNCH_CHECK("gap_partial");
  hipLaunchKernelGGL(gap_final_kernel, dim3(ceil_div(BC, 64)), dim3(64), 0, st, scratch, cond, BC, nsplit, hw);
  HRSEG_LAUNCH_CHECK("gap_final");
  return 0;
}

extern "C" int hrseg_film_linear_fwd(const float* cond, const float* wl, const float* bl, float* gb, int B, int Cc,
                                     int F2, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(cond && wl && bl && gb && B > 0 && Cc > 0 && F2 > 0, "hrseg_film_linear_fwd: bad arguments");
  hipLaunchKernelGGL(film_linear_fwd_kernel, dim3(ceil_div((long)B * F2, 256)), dim3(256), 0, (hipStream_t)stream, cond,
                     wl, bl, gb, B, Cc, F2);
  HRSEG_LAUNCH_CHECK("film_linear_fwd");
  return 0;
}

extern "C" int hrseg_film_linear_bwd(const float* cond, const float* wl, const float* dgb, float* dcond, float* dwl,
                                     float* dbl, int B, int Cc, int F2, float dcond_scale, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(cond && wl && dgb && B > 0 && Cc > 0 && F2 > 0, "hrseg_film_linear_bwd: bad arguments");
  hipLaunchKernelGGL(film_linear_bwd_kernel, dim3(ceil_div(F2, 256)), dim3(256), 0, (hipStream_t)stream, cond, wl, dgb,
                     dcond, dwl, dbl, B, Cc, F2, dcond_scale);
  HRSEG_LAUNCH_CHECK("film_linear_bwd");
  return 0;
}

extern "C" int hrseg_head_fwd(const float* f, int ldf, const float* gb, const float* w, const float* bias, float* z,
                              int ldz, int B, long hw, int F, int Cout, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(f && w && z && B > 0 && hw > 0 && F > 0 && F % 4 == 0 && Cout > 0 && Cout <= 8 && ldf >= F &&
                      ldf % 4 == 0 && ldz >= Cout,
                  "hrseg_head_fwd: bad arguments (F=%d Cout=%d)", F, Cout);
  hipStream_t st = (hipStream_t)stream;
  const int co = Cout <= 4 ? 4 : 8;
  const size_t smem = (size_t)(co * F + co) * sizeof(float);
  const int lp = (F / 4 <= 16) ? 16 : 64;
  long blocks = (hw + (256 / lp) - 1) / (256 / lp);
  const long cap = (2048 + B - 1) / B < 64 ? 64 : (2048 + B - 1) / B;      // ~2048 blocks in all: the per-block set-up is paid once per CU slot
  if (blocks > cap) blocks = cap;
  dim3 grid((int)blocks, B);
#define HF(LP_, CO_) hipLaunchKernelGGL((head_fwd_kernel<LP_, CO_>), grid, dim3(256), smem, st, f, ldf, gb, w, bias, z, ldz, hw, F, Cout)
  if (lp == 16 && co == 4) HF(16, 4);
  else if (lp == 16) HF(16, 8);
  else if (co == 4) HF(64, 4);
  else HF(64, 8);
#undef HF
  HRSEG_LAUNCH_CHECK("head_fwd");
  return 0;
}

extern "C" int hrseg_head_bwd(const float* f, int ldf, const float* gb, const float* w, const float* dz, int lddz,
                              float* df, int lddf, int df_accumulate, float* dw, float* dbias, float* dgb, int B,
                              long hw, int F, int Cout, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(f && w && dz && dw && B > 0 && hw > 0 && F > 0 && F % 4 == 0 && F <= 1024 && Cout > 0 &&
                      Cout <= 8 && (gb == nullptr) == (dgb == nullptr),
                  "hrseg_head_bwd: bad arguments (F=%d Cout=%d)", F, Cout);
  hipStream_t st = (hipStream_t)stream;
  long chunks = 2048 / B;
  if (chunks < 1) chunks = 1;
  long ppb = (hw + chunks - 1) / chunks;
  if (ppb < 64) ppb = 64;
  // deterministic: one block per image, the images one launch after the other (a single adder per dW element at a time)
  const int nlaunch = hrseg_g_deterministic ? B : 1;
  if (hrseg_g_deterministic) ppb = hw;
  dim3 grid(ceil_div(hw, ppb), hrseg_g_deterministic ? 1 : B);
  for (int b0 = 0; b0 < nlaunch; ++b0) {
    if (Cout <= 4)
      hipLaunchKernelGGL((head_bwd_kernel<4>), grid, dim3(256), 0, st, f, ldf, gb, w, dz, lddz, df, lddf, df_accumulate,
                         dw, dbias, dgb, hw, F, Cout, ppb, b0);
    else
      hipLaunchKernelGGL((head_bwd_kernel<8>), grid, dim3(256), 0, st, f, ldf, gb, w, dz, lddz, df, lddf, df_accumulate,
                         dw, dbias, dgb, hw, F, Cout, ppb, b0);
  }
  HRSEG_LAUNCH_CHECK("head_bwd");
  return 0;
}

static float up_scale(int in_size, int out_size, int align) {
  if (align) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  return (float)in_size / (float)out_size;
}

extern "C" int hrseg_logits_up_fwd(const float* in, int ldin, int B, int Hi, int Wi, int C, float* out, int Ho, int Wo,
                                   int align_corners, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(in && out && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C <= MAXC && ldin >= C,
                  "hrseg_logits_up_fwd: bad arguments");
  const long n = (long)B * Ho * Wo;
  hipLaunchKernelGGL(logits_up_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, in, ldin, B, Hi,
                     Wi, C, out, Ho, Wo, up_scale(Hi, Ho, align_corners), up_scale(Wi, Wo, align_corners),
                     align_corners);
  HRSEG_LAUNCH_CHECK("logits_up_fwd");
  return 0;
}

extern "C" int hrseg_logits_up_bwd(const float* dout, int B, int Hi, int Wi, int C, float* din, int lddin, int Ho,
                                   int Wo, int align_corners, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(dout && din && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C <= MAXC && lddin >= C,
                  "hrseg_logits_up_bwd: bad arguments");
  const long n = (long)B * Hi * Wi;
  hipLaunchKernelGGL(logits_up_bwd_kernel, dim3(ceil_div(n, 128)), dim3(128), 0, (hipStream_t)stream, dout, B, Hi, Wi,
                     C, din, lddin, Ho, Wo, up_scale(Hi, Ho, align_corners), up_scale(Wi, Wo, align_corners),
                     align_corners);
  HRSEG_LAUNCH_CHECK("logits_up_bwd");
  return 0;
}

extern "C" int hrseg_sigmoid_fwd(const float* z, float* p, long n, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(z && p && n > 0, "hrseg_sigmoid_fwd: bad arguments");
  hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, z, p, n);
  HRSEG_LAUNCH_CHECK("sigmoid_fwd");
  return 0;
}

extern "C" int hrseg_sigmoid_bwd(const float* dp, long sb, long sc, long si, const float* z, float* dz, int accumulate,
                                 int B, int C, long hw, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(dp && z && dz && B > 0 && C > 0 && hw > 0, "hrseg_sigmoid_bwd: bad arguments");
  const long n = (long)B * C * hw;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, dp, sb, sc, si, z,
                     dz, accumulate, C, hw, n);
  HRSEG_LAUNCH_CHECK("sigmoid_bwd");
  return 0;
}

extern "C" int hrseg_compose_fwd(const float* z, const float* pprev, float* p, int B, int C, int Cprev, long hw,
                                 int ngroups, const int* group_parent, const int* group_size, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(z && pprev && p && B > 0 && C > 0 && C <= MAXC && Cprev > 0 && Cprev <= MAXC && hw > 0 && ngroups > 0,
                  "hrseg_compose_fwd: bad arguments");
  Groups g;
  if (int e = fill_groups(g, ngroups, group_parent, group_size, C, Cprev, "hrseg_compose_fwd")) return e;
  const long n = (long)B * hw;
  hipLaunchKernelGGL(compose_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, z, pprev, p, C,
                     Cprev, hw, n, g);
  HRSEG_LAUNCH_CHECK("compose_fwd");
  return 0;
}

extern "C" int hrseg_compose_bwd(const float* dp, long sb, long sc, long si, const float* z, const float* pprev,
                                 float* dz, int dz_accumulate, float* dpprev, int dpprev_accumulate, int B, int C,
                                 int Cprev, long hw, int ngroups, const int* group_parent, const int* group_size,
                                 hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(dp && z && pprev && B > 0 && C > 0 && C <= MAXC && Cprev > 0 && Cprev <= MAXC && hw > 0 && ngroups > 0,
                  "hrseg_compose_bwd: bad arguments");
  Groups g;
  if (int e = fill_groups(g, ngroups, group_parent, group_size, C, Cprev, "hrseg_compose_bwd")) return e;
  const long n = (long)B * hw;
  hipLaunchKernelGGL(compose_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, dp, sb, sc, si, z,
                     pprev, dz, dz_accumulate, dpprev, dpprev_accumulate, C, Cprev, hw, n, g);
  HRSEG_LAUNCH_CHECK("compose_bwd");
  return 0;
}

extern "C" int hrseg_loss_partials(const float* z, const float* t, double* partial, int B, int C, long hw,
                                   hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(z && t && partial && B > 0 && C > 0 && C <= MAXC && hw > 0, "hrseg_loss_partials: bad arguments (C=%d)", C);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_f64_kernel, dim3(1), dim3(256), 0, st, partial, B * C * 5);   // kernel node, not memset
  long chunks = hrseg_g_deterministic ? 1 : 1024 / B;      // deterministic: one block (one adder) per image
  if (chunks < 1) chunks = 1;
  long ppb = (hw + chunks - 1) / chunks;
  if (ppb < 256) ppb = 256;
  dim3 grid(ceil_div(hw, ppb), B);
  if (C <= 4) hipLaunchKernelGGL((loss_partials_kernel<4>), grid, dim3(256), 0, st, z, t, partial, C, hw, ppb);
  else if (C <= 8) hipLaunchKernelGGL((loss_partials_kernel<8>), grid, dim3(256), 0, st, z, t, partial, C, hw, ppb);
  else hipLaunchKernelGGL((loss_partials_kernel<16>), grid, dim3(256), 0, st, z, t, partial, C, hw, ppb);
  HRSEG_LAUNCH_CHECK("loss_partials");
  return 0;
}

extern "C" int hrseg_loss_finalize(const double* partial, const float* w, int B, int C, float* out, float* coef,
                                   hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(partial && w && out && coef && B > 0 && C > 0, "hrseg_loss_finalize: bad arguments");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, w, B, C, out, coef);
  HRSEG_LAUNCH_CHECK("loss_finalize");
  return 0;
}

extern "C" int hrseg_loss_bwd(const float* z, const float* t, const float* coef, const float* g, float* dz,
                              int accumulate, int B, int C, long hw, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(z && t && coef && g && dz && B > 0 && C > 0 && C <= MAXC && hw > 0, "hrseg_loss_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(hw, 256), B);
  if (C <= 4) hipLaunchKernelGGL((loss_bwd_kernel<4>), grid, dim3(256), 0, st, z, t, coef, g, dz, accumulate, C, hw);
  else if (C <= 8) hipLaunchKernelGGL((loss_bwd_kernel<8>), grid, dim3(256), 0, st, z, t, coef, g, dz, accumulate, C, hw);
  else hipLaunchKernelGGL((loss_bwd_kernel<16>), grid, dim3(256), 0, st, z, t, coef, g, dz, accumulate, C, hw);
  HRSEG_LAUNCH_CHECK("loss_bwd");
  return 0;
}

extern "C" int hrseg_consistency(const float* p, const float* pprev, double* out, int B, int C, int Cprev, long hw,
                                 int ngroups, const int* group_parent, const int* group_size, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(p && pprev && out && B > 0 && C > 0 && C <= MAXC && Cprev > 0 && Cprev <= MAXC && hw > 0 && ngroups > 0,
                  "hrseg_consistency: bad arguments");
  Groups g;
  if (int e = fill_groups(g, ngroups, group_parent, group_size, C, Cprev, "hrseg_consistency")) return e;
  const long n = (long)B * hw;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (hrseg_g_deterministic) blocks = 1;
  hipLaunchKernelGGL(consistency_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, pprev, out, C, Cprev,
                     hw, n, g);
  HRSEG_LAUNCH_CHECK("consistency");
  return 0;
}

extern "C" int hrseg_predict_metrics(const float* z, const float* t, float* onehot, long long* cm, int B, int C,
                                     long hw, int child, int mask_pred, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(z && t && cm && B > 0 && C > 0 && C <= MAXC && hw > 0, "hrseg_predict_metrics: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * hw;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  dim3 grid((int)blocks);
  unsigned long long* c = (unsigned long long*)cm;
  if (C <= 4) hipLaunchKernelGGL((predict_metrics_kernel<4>), grid, dim3(256), 0, st, z, t, onehot, c, C, hw, n, child, mask_pred);
  else if (C <= 8) hipLaunchKernelGGL((predict_metrics_kernel<8>), grid, dim3(256), 0, st, z, t, onehot, c, C, hw, n, child, mask_pred);
  else hipLaunchKernelGGL((predict_metrics_kernel<16>), grid, dim3(256), 0, st, z, t, onehot, c, C, hw, n, child, mask_pred);
  HRSEG_LAUNCH_CHECK("predict_metrics");
  return 0;
}

// per-class metric vectors of up to 8 levels from their confusion counts, one launch (Metrics/performance_metrics.py
// metrics_from_confusion: the reference's IoU / Dice / precision / recall classes, train.py:47-51, evaluated in fp64 on the
// counts and rounded to fp32; a zero denominator gives 0).  Thread = one (level, class).
#define METRIC_MAXL 8
struct MetricLevels { int n, total; const long long* cm[METRIC_MAXL]; int K[METRIC_MAXL], child[METRIC_MAXL], off[METRIC_MAXL]; };
__global__ void metric_vectors_kernel(MetricLevels a, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.total) return;
  int L = 0;
  while (L + 1 < a.n && i >= a.off[L + 1]) ++L;
  const int K = a.K[L], ch = a.child[L], c = i - a.off[L] + ch;        // c: label of this class in the K x K matrix
  const long long* cm = a.cm[L];
  // child levels: pixels whose TARGET is the synthetic background label 0 are dropped (rows 1..K-1 only)
  double tp = (double)cm[(long)c * K + c], row = 0.0, col = 0.0;
  for (int j = 0; j < K; ++j) row += (double)cm[(long)c * K + j];
  for (int r = ch; r < K; ++r) col += (double)cm[(long)r * K + c];
  const double fn = row - tp, fp = col - tp;
  auto sdiv = [](double x, double y) { return y == 0.0 ? 0.f : (float)(x / y); };
  const float rec = sdiv(tp, tp + fn);
  out[0 * a.total + i] = rec;                                  // "accuracy" (per-class accuracy = recall, as in the reference)
  out[1 * a.total + i] = sdiv(tp, tp + fp + fn);               // iou
  out[2 * a.total + i] = sdiv(2.0 * tp, 2.0 * tp + fp + fn);   // dice
  out[3 * a.total + i] = sdiv(tp, tp + fp);                    // precision
  out[4 * a.total + i] = rec;                                  // recall
}
extern "C" int hrseg_metric_vectors(int nlevels, const long long* const* cm, const int* K, const int* child, float* out,
                                    hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(nlevels >= 1 && nlevels <= METRIC_MAXL && cm && K && child && out, "hrseg_metric_vectors: 1..%d levels", METRIC_MAXL);
  MetricLevels a;
  a.n = nlevels;
  a.total = 0;
  for (int L = 0; L < nlevels; ++L) {
    HRSEG_CHECK_ARG(cm[L] && K[L] > (child[L] ? 1 : 0) && K[L] <= MAXC + 1, "hrseg_metric_vectors: level %d: bad matrix", L);
    a.cm[L] = cm[L];
    a.K[L] = K[L];
    a.child[L] = child[L] ? 1 : 0;
    a.off[L] = a.total;
    a.total += K[L] - a.child[L];
  }
  hipLaunchKernelGGL(metric_vectors_kernel, dim3((a.total + 63) / 64), dim3(64), 0, (hipStream_t)stream, a, out);
  HRSEG_LAUNCH_CHECK("metric_vectors");
  return 0;
}

extern "C" int hrseg_group_kl(const float* z, const float* pprev, double* out, int B, int C, int Cprev, long hw,
                              int ngroups, const int* group_parent, const int* group_size, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(z && pprev && out && B > 0 && C > 0 && C <= MAXC && Cprev > 0 && Cprev <= MAXC && hw > 0 && ngroups > 0,
                  "hrseg_group_kl: bad arguments");
  Groups g;
  if (int e = fill_groups(g, ngroups, group_parent, group_size, C, Cprev, "hrseg_group_kl")) return e;
  const long n = (long)B * hw;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (hrseg_g_deterministic) blocks = 1;
  hipLaunchKernelGGL(group_kl_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, z, pprev, out, C, Cprev, hw, n, g);
  HRSEG_LAUNCH_CHECK("group_kl");
  return 0;
}

extern "C" int hrseg_group_kl_bwd(const float* z, const float* pprev, const float* g, float scale, float* dz, int B, int C,
                                  int Cprev, long hw, int ngroups, const int* group_parent, const int* group_size,
                                  hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(z && pprev && g && dz && B > 0 && C > 0 && C <= MAXC && Cprev > 0 && Cprev <= MAXC && hw > 0 && ngroups > 0,
                  "hrseg_group_kl_bwd: bad arguments");
  Groups gr;
  if (int e = fill_groups(gr, ngroups, group_parent, group_size, C, Cprev, "hrseg_group_kl_bwd")) return e;
  const long n = (long)B * hw;
  hipLaunchKernelGGL(group_kl_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, z, pprev, g, scale, dz,
                     C, Cprev, hw, n, gr);
  HRSEG_LAUNCH_CHECK("group_kl_bwd");
  return 0;
}

extern "C" int hrseg_consistency_bwd(const float* p, const float* pprev, const float* g, float scale, float* dp,
                                     float* dpprev, int B, int C, int Cprev, long hw, int ngroups,
                                     const int* group_parent, const int* group_size, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(p && pprev && g && dp && dpprev && B > 0 && C > 0 && C <= MAXC && Cprev > 0 && Cprev <= MAXC &&
                      hw > 0 && ngroups > 0,
                  "hrseg_consistency_bwd: bad arguments");
  Groups gr;
  if (int e = fill_groups(gr, ngroups, group_parent, group_size, C, Cprev, "hrseg_consistency_bwd")) return e;
  const long n = (long)B * hw;
  hipLaunchKernelGGL(consistency_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, p, pprev, g,
                     scale, dp, dpprev, C, Cprev, hw, n, gr);
  HRSEG_LAUNCH_CHECK("consistency_bwd");
  return 0;
}

// --------------------------------------------------------------------------- target encoding
// Label image (one pixel value per leaf class) -> the per-node target planes the losses and
// metrics consume: SegDataset.separate_masks / traverse_tree / process_ignore_values of the reference
// (Data/dataset.py:41-124, 227-265) for an identity spatial transform.  on_lut[v] has bit c set when
// channel c's node contains the leaf whose pixel value is v (a parent = OR of its leaves), so one
// 8-byte table read per pixel replaces the per-node mask images.  HBM-bound: 1 byte in, 4*C out.
struct EncodeParents { int parent[64]; };
__global__ __launch_bounds__(256) void encode_targets_kernel(const unsigned char* __restrict__ label,
                                                             const unsigned long long* __restrict__ on_lut,
                                                             EncodeParents pr, float* __restrict__ out, int C,
                                                             long hw, long n) {
  __shared__ unsigned long long lut[256];
  lut[threadIdx.x] = on_lut[threadIdx.x];
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / hw, pix = i - b * hw;
    const unsigned long long bits = lut[label[i]];
    float* o = out + (size_t)b * C * hw + pix;
    for (int c = 0; c < C; ++c) {
      const int par = pr.parent[c];
      float v;
      if ((bits >> c) & 1ull) v = 1.f;
      else if (par < 0 || ((bits >> par) & 1ull)) v = 0.f;   // root, or inside the direct parent's area
      else v = -1.f;                                          // outside the parent: ignored by loss and metrics
      o[(size_t)c * hw] = v;
    }
  }
}

extern "C" int hrseg_encode_targets(const unsigned char* label, const unsigned long long* on_lut,
                                    const int* parent, float* out, int B, int C, long hw, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(label && on_lut && parent && out && B > 0 && hw > 0, "hrseg_encode_targets: bad arguments");
  HRSEG_CHECK_ARG(C >= 1 && C <= 64, "hrseg_encode_targets: C=%d not in 1..64", C);
  EncodeParents pr;
  for (int c = 0; c < 64; ++c) pr.parent[c] = -1;
  for (int c = 0; c < C; ++c) {
    HRSEG_CHECK_ARG(parent[c] >= -1 && parent[c] < C, "hrseg_encode_targets: parent[%d]=%d out of range", c, parent[c]);
    pr.parent[c] = parent[c];
  }
  const long n = (long)B * hw;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(encode_targets_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, label, on_lut, pr,
                     out, C, hw, n);
  HRSEG_LAUNCH_CHECK("encode_targets");
  return 0;
}

// --------------------------------------------------------------------------- level synthesis for flat models
// predictEval.py:85-185 (get_parent_masks / combine_levels): out[b][o][pix] is, per output channel o, either the COPY
// of one input channel (mask with a single bit) or the UNION "any selected channel > 0" (1.0 / 0.0) of the input
// channels whose bits are set.  Input channel i < C0 is plane i of x0, else plane i - C0 of x1 (the leaves and the
// already synthesised parents of combine_levels).  One thread per pixel, planes coalesced.
struct LevelTable { unsigned long long mask[64]; int is_union[64]; };
__global__ __launch_bounds__(256) void combine_levels_kernel(const float* __restrict__ x0, int C0,
                                                             const float* __restrict__ x1, int C1,
                                                             float* __restrict__ out, int Cout, long hw, long n,
                                                             LevelTable tab) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / hw, pix = i - b * hw;
    for (int o = 0; o < Cout; ++o) {
      unsigned long long m = tab.mask[o];
      float v = 0.f;
      while (m) {
        const int c = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float xv = (c < C0) ? x0[((size_t)b * C0 + c) * hw + pix] : x1[((size_t)b * C1 + (c - C0)) * hw + pix];
        if (tab.is_union[o]) v = (xv > 0.f) ? 1.f : v;
        else v = xv;
      }
      out[((size_t)b * Cout + o) * hw + pix] = v;
    }
  }
}

extern "C" int hrseg_combine_levels(const float* x0, int C0, const float* x1, int C1, const unsigned long long* masks,
                                    const int* is_union, float* out, int B, int Cout, long hw, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(x0 && masks && is_union && out && B > 0 && hw > 0 && C0 > 0 && C1 >= 0 && C0 + C1 <= 64 && Cout > 0 && Cout <= 64 &&
                      (C1 == 0 || x1), "hrseg_combine_levels: bad arguments (C0=%d C1=%d Cout=%d)", C0, C1, Cout);
  LevelTable tab;
  for (int o = 0; o < Cout; ++o) {
    HRSEG_CHECK_ARG(masks[o] != 0 && (C0 + C1 == 64 || (masks[o] >> (C0 + C1)) == 0),
                    "hrseg_combine_levels: channel %d selects no input / an input beyond %d", o, C0 + C1);
    tab.mask[o] = masks[o];
    tab.is_union[o] = is_union[o];
  }
  const long n = (long)B * hw;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(combine_levels_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x0, C0, x1, C1, out, Cout,
                     hw, n, tab);
  HRSEG_LAUNCH_CHECK("combine_levels");
  return 0;
}
