// Split-precision implicit GEMM on the bf16 matrix pipe (gfx950), included by conv.hip.
//
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 value splits EXACTLY into three bf16
// pieces (8 significand bits each: hi = truncate(x), mid = truncate(x - hi), lo = x - hi - mid), so an fp32
// product is the sum of nine bf16 products of which the three smallest (mid*lo, lo*mid, lo*lo, each below
// 2^-24 of the full product) are dropped: six v_mfma_f32_16x16x32_bf16 with fp32 accumulation per tile and
// K step of 32 do the work of eight fp32 MFMAs in 6/16 of their time ("bf16x3", NS = 3: fp32-grade results).
// NS = 2 keeps two pieces / three products (operand error 2^-16), NS = 1 is plain bf16 inputs with fp32
// accumulation (BASELINE configs[4] arithmetic).  Activations and weights stay fp32 in HBM; the split
// happens on the fly (v_perm_b32 packs two truncated values, v_and + v_sub form the residual).
//
// Measured on MI355X (tools/ubench/bf16_rate.hip): split + MFMA loop, two waves per SIMD: NS=3 298-323 TFLOP/s
// fp32-equivalent (1.8-1.9 PFLOP/s on the matrix pipe), NS=2 535-562, NS=1 1355; the K=16 form
// v_mfma_f32_16x16x16_bf16 runs at HALF the FLOP rate of 16x16x32, so only the K=32 form is used and the
// reduction index runs over 32-wide slabs of the flattened (tap, 16-channel chunk) list.
//
// Layout of one block (256 threads, 4 waves): GEMM rows = 64*WTM output pixels, each wave owns 16*WTM of
// them; columns = 16*WTN output channels.  The PIXEL operand never touches LDS: lane (r = l&15, g = l>>4)
// loads, for its own pixel row r of every 16-row tile, channels 4g..4g+3 of the slab's two 16-channel units
// (two 16-byte buffer loads), splits them in registers and has the MFMA B fragment (k = 8g..8g+7) in place.
// Only the WEIGHT slab goes through LDS (all four waves read every weight fragment): [piece][row][64 B],
// XOR-swizzled 16-byte slots, double buffered, one barrier per slab.  The k index inside a slab is
// permuted the same way on both operands: k = 8g+j  <->  unit j>>2, channel 4g + (j&3).
#pragma once
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// two floats -> one dword of two bf16 (element 0 in the low half): truncation (exact residual arithmetic)
__device__ __forceinline__ unsigned sp_pack_trunc(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// ... round to nearest even (the last piece when fewer than three pieces are kept)
__device__ __forceinline__ unsigned sp_pack_rne(float lo, float hi) {
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float sp_trunc(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

// Scheme codes (template parameter NS): 1, 2, 3 = that many bf16 pieces; 4 = "fp16x2": two fp16 pieces (11
// significand bits each: hi = round-toward-zero(x), lo = round-to-nearest(x - hi), 22 bits in all) and the three
// products hi*hi, hi*lo, lo*hi on v_mfma_f32_16x16x32_f16 -- half the matrix work of bf16x3 at 2^-22 operand
// error.  fp16 has 5 exponent bits, so fp16x2 operands are SCALED by a power of two before the split (weights by a
// fixed 2^8, gradients by 2^14 / their measured |max|, see sp_pow2_scale) and the accumulators are scaled back in
// the epilogue; activations are neither scaled nor clamped (exact to 22 bits up to |x| = 65504, Inf / NaN beyond ~1.3e5:
// include/hrseg.h; ops.py range-checks them in deterministic mode).  Elements more than 2^19 below the scaled maximum
// lose their low piece to the subnormal range (absolute error <= 2^-25 after scaling): invisible in a dot product.
constexpr int sp_np(int ns) { return ns == 4 ? 2 : ns; }                       // pieces per operand
constexpr int sp_products(int ns) { return ns == 1 ? 1 : ns == 3 ? 6 : 3; }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// power-of-two scale that brings the tensor's |max| into [2^14, 2^15) and its inverse (1, 1 for 0 / denormal /
// absent).  `absmax` is the 64-slot array hrseg_bn_bwd_group fills (each slot the max over a share of the blocks).
__device__ __forceinline__ void sp_pow2_scale(const float* absmax, float& scale, float& inv) {
  scale = inv = 1.f;
  if (!absmax) return;
  float m = absmax[threadIdx.x & 63];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const unsigned e = (__float_as_uint(m) >> 23) & 255u;
  if (e >= 16u && e <= 250u) {
    scale = __uint_as_float((268u - e) << 23);
    inv = __uint_as_float((e - 14u) << 23);
  }
}

__device__ __forceinline__ unsigned sp_pack_f16_rtz(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}
__device__ __forceinline__ unsigned sp_pack_f16_rne(float lo, float hi) {
  const f16x2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float sp_f16_lo(unsigned pk) { return (float)__builtin_bit_cast(f16x2, pk)[0]; }
__device__ __forceinline__ float sp_f16_hi(unsigned pk) { return (float)__builtin_bit_cast(f16x2, pk)[1]; }

// N fp32 (N = 4 or 8, N/2 dwords per piece) -> sp_np(NS) pieces; `sc` scales first (fp16x2 only)
template <int NS, int N>
__device__ __forceinline__ void sp_split(float (&x)[N], unsigned (&out)[sp_np(NS)][N / 2], float sc) {
  if (NS == 4) {
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] *= sc;      // no clamp: NaN propagates, |x| beyond fp16 range ends in Inf / NaN (loud), see hrseg.h
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
      const unsigned hi = sp_pack_f16_rtz(x[2 * j], x[2 * j + 1]);
      out[0][j] = hi;
      out[1][j] = sp_pack_f16_rne(x[2 * j] - sp_f16_lo(hi), x[2 * j + 1] - sp_f16_hi(hi));
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < sp_np(NS); ++s) {
    const bool last = s == NS - 1;
#pragma unroll
    for (int j = 0; j < N / 2; ++j)
      out[s][j] = (last && NS < 3) ? sp_pack_rne(x[2 * j], x[2 * j + 1]) : sp_pack_trunc(x[2 * j], x[2 * j + 1]);
    if (!last) {
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] -= sp_trunc(x[j]);
    }
  }
}
// 8 fp32 (two f32x4: k = 0..3 and 4..7 of this lane's fragment) -> fragments (128-bit, typed bf16x8 whatever the scheme)
template <int NS>
__device__ __forceinline__ void sp_split8(const f32x4& a, const f32x4& b, bf16x8 (&out)[sp_np(NS)], float sc = 1.f) {
  float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  unsigned u[sp_np(NS)][4];
  sp_split<NS, 8>(x, u, sc);
#pragma unroll
  for (int s = 0; s < sp_np(NS); ++s) out[s] = __builtin_bit_cast(bf16x8, (u32x4){u[s][0], u[s][1], u[s][2], u[s][3]});
}
// 4 fp32 -> pieces of 4 elements (8 bytes each): the staging granule
template <int NS>
__device__ __forceinline__ void sp_split4(const f32x4& a, u32x2 (&out)[sp_np(NS)], float sc = 1.f) {
  float x[4] = {a[0], a[1], a[2], a[3]};
  unsigned u[sp_np(NS)][2];
  sp_split<NS, 4>(x, u, sc);
#pragma unroll
  for (int s = 0; s < sp_np(NS); ++s) out[s] = u32x2{u[s][0], u[s][1]};
}

// acc += W-fragment pieces x X-fragment pieces: the products whose weight is at least 2^-16 (bf16) / 2^-11 (fp16)
// of the full product
template <int NS>
__device__ __forceinline__ f32x4 sp_mma(const bf16x8 (&w)[sp_np(NS)], const bf16x8 (&x)[sp_np(NS)], f32x4 acc) {
  if (NS == 4) {
    const f16x8 w0 = __builtin_bit_cast(f16x8, w[0]), w1 = __builtin_bit_cast(f16x8, w[1]);
    const f16x8 x0 = __builtin_bit_cast(f16x8, x[0]), x1 = __builtin_bit_cast(f16x8, x[1]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x1, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0, acc, 0, 0, 0);
  }
  if (NS == 3) {   // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[sp_np(NS) - 1], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[sp_np(NS) - 1], acc, 0, 0, 0);
  }
  if (NS >= 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1 % sp_np(NS)], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1 % sp_np(NS)], acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], acc, 0, 0, 0);
}

// One product of sp_mma (pr = 0 .. sp_nprod(NS)-1, in sp_mma's order).  The kernels issue the products of a slab
// product-outermost -- all accumulators' first product, then all second ones ... -- so that two MFMAs on the same
// accumulator are never back to back: a dependent 16x16x32 MFMA waits for its predecessor's full latency (twice its
// issue time), and the compiler keeps source order inside an unrolled slab.  Every accumulator still receives its
// products in sp_mma's order, so results are bit-identical to the chained form.
__host__ __device__ constexpr int sp_nprod(int ns) { return ns == 3 ? 6 : ns == 1 ? 1 : 3; }
template <int NS>
__device__ __forceinline__ f32x4 sp_mma_p(int pr, const bf16x8 (&w)[sp_np(NS)], const bf16x8 (&x)[sp_np(NS)], f32x4 acc) {
  int wi = 0, xi = 0;
  if (NS == 3) {
    wi = (pr == 0 || pr == 3) ? 1 : (pr == 1) ? 2 : 0;
    xi = (pr == 0 || pr == 4) ? 1 : (pr == 2) ? 2 : 0;
  } else if (NS != 1) {
    wi = pr == 0 ? 1 : 0;
    xi = pr == 1 ? 1 : 0;
  }
  if (NS == 4)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[wi]), __builtin_bit_cast(f16x8, x[xi]), acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[wi], x[xi], acc, 0, 0, 0);
}

template <int NS, int WTN>
struct SpLds {
  static constexpr int BN = 16 * WTN;
  static constexpr int PIECE = BN * 64;          // bytes: one slab of one piece, [BN rows][32 bf16]
  static constexpr int STAGE = sp_np(NS) * PIECE;       // one buffer
  static constexpr int BYTES = 2 * STAGE;        // double buffered
};

#define SP_DEPTH 3        // slabs of global-load look-ahead in igemm_sp_body
// One output tile of one convolution.  `ks_idx / ks_n`: split-K slice of the slab list (partial sums are
// added with fp32 atomics, as in igemm_body).
template <int NS, int WTM, int WTN>
__device__ __forceinline__ void igemm_sp_body(const IgemmArgs& p, unsigned char* lds, const int bid, const int nblk,
                                              const int ks_idx, const int ks_n) {
  constexpr int BM = 64 * WTM, BN = 16 * WTN;
  constexpr int PIECE = SpLds<NS, WTN>::PIECE, STAGE = SpLds<NS, WTN>::STAGE;
  constexpr int WG = BN * 8;                       // 16-byte weight granules per slab
  constexpr int W_LOADS = (WG + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = p.N / BN;
  const int wg = xcd_remap(bid, nblk);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;

  float xscale, xinv;                       // fp16x2: power-of-two scale of the pixel operand (a gradient: from its |max|)
  sp_pow2_scale(p.xmax, xscale, xinv);
  const float oscale = xinv * p.wscale_inv;

  // reduction index: units of 16 channels, u = tap * kch + chunk; a slab = units 2s, 2s+1
  const int kch = p.K >> 4;
  const int nunits = p.ntaps * kch;
  const int nslabs_all = (nunits + 1) >> 1;
  const int per = (nslabs_all + ks_n - 1) / ks_n;
  const int s_lo = ks_idx * per;
  const int s_hi = min(s_lo + per, nslabs_all);
  const int nslabs = s_hi - s_lo;

  // Input descriptor: based at the first image this tile touches, moved back by the most negative tap
  // offset, so that every lane offset and every per-tap scalar offset is non-negative.  Memory in front of
  // the tensor is never read: taps outside the image get the out-of-range offset (zero fill).
  const int hw = p.Ho * p.Wo;
  const int b0 = m0 / hw;
  const long tap0 = (long)p.oy_min * p.Wi + p.ox_min;          // <= 0
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + ((long)b0 * p.Hi * p.Wi + tap0) * p.ldx,
                                              (size_t)((long)(p.B - b0) * p.Hi * p.Wi - tap0) * p.ldx * 4);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (size_t)p.N * p.T * p.K * 4);

  // this lane's pixel rows: byte offset of (pixel at tap offset 0, channel 4g) and the taps that fall outside
  unsigned voff[WTM];
  int inval[WTM];
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + r16;
    if (row < p.M) {
      const int b = fdiv(row, hw, p.rcp_hw);
      const int rem = row - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      const int iy0 = oy * p.sy, ix0 = ox * p.sx;
      voff[m] = ((unsigned)((b - b0) * p.Hi * p.Wi + iy0 * p.Wi + ix0) * (unsigned)p.ldx + 4u * g) * 4u;
      int bad = 0;
      for (int t = 0; t < p.ntaps; ++t) {
        const int iy = iy0 + (int)((p.offy_pk >> (4 * t)) & 15) - 8, ix = ix0 + (int)((p.offx_pk >> (4 * t)) & 15) - 8;
        bad |= ((iy < 0) | (iy >= p.Hi) | (ix < 0) | (ix >= p.Wi)) ? (1 << t) : 0;
      }
      inval[m] = bad;
    } else {
      voff[m] = 0;
      inval[m] = -1;
    }
  }

  // weight granules of this thread: granule f = (row n = f>>3, unit (f>>2)&1, 4-channel group f&3)
  unsigned wbase[W_LOADS];
  int wunit[W_LOADS], wst[W_LOADS];
#pragma unroll
  for (int i = 0; i < W_LOADS; ++i) {
    const int f = tid + 256 * i;
    const int n = f >> 3, unit = (f >> 2) & 1, gq = f & 3;
    wbase[i] = (f < WG) ? ((unsigned)(n0 + n) * (unsigned)(p.T * p.K) + 4u * gq) * 4u : HRSEG_BUF_OOB;
    wunit[i] = unit;
    wst[i] = n * 64 + lds_slot(n, gq) * 16 + unit * 8;
  }

  // running unit counters (scalar): tap and chunk of the next slab's two units.  Loads run SP_DEPTH slabs
  // ahead of their use in a ring of register sets: a slab of this body is 6-36 MFMAs per wave (0.1-0.3 us), an
  // L2 round trip under load is several times that, and with one or two blocks per CU nothing else hides it
  // (measured: 1.2 us per slab with one slab of look-ahead on the low-resolution branches).  Every load is
  // issued unconditionally -- past the end of the slice with out-of-range offsets (zero fill, no traffic) -- so
  // the vmcnt bookkeeping stays exact and a slab waits for its own loads only.
  constexpr int D = SP_DEPTH;
  int u_next = 2 * s_lo;
  const int u_end = min(nunits, 2 * s_hi);
  // (tap, chunk) of unit u_next, stepped unit by unit: ONE division per block (a division per unit is ~25 vector instructions
  // twice per slab, in a body whose slab is 6-36 MFMAs)
  int u_t = __builtin_amdgcn_readfirstlane(u_next / kch), u_c = u_next - u_t * kch;
  f32x4 ra[D][WTM][2], rwt[D][W_LOADS];
  auto issue_loads = [&](f32x4 (&ra)[WTM][2], f32x4 (&rwt)[W_LOADS]) {
    unsigned soff[2], wsoff[2];
    int tapbit[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int u = u_next + h;
      const bool live = u < u_end;
      const int t = live ? u_t : 0;
      const int c = live ? u_c : 0;
      if (++u_c == kch) { u_c = 0; ++u_t; }
      const int dy = (int)((p.offy_pk >> (4 * t)) & 15) - 8 - p.oy_min, dx = (int)((p.offx_pk >> (4 * t)) & 15) - 8 - p.ox_min;
      soff[h] = (unsigned)((dy * p.Wi + dx) * p.ldx + 16 * c) * 4u;
      wsoff[h] = live ? (unsigned)((int)((p.wtap_pk >> (4 * t)) & 15) * p.K + 16 * c) * 4u : HRSEG_BUF_OOB;
      tapbit[h] = live ? t : 31;          // bit 31 of inval is set only for rows past M; dead unit: forced below
      if (!live) soff[h] = 0;
    }
    const bool live0 = u_next < u_end, live1 = u_next + 1 < u_end;
#pragma unroll
    for (int m = 0; m < WTM; ++m) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // -1 (out of range: zero fill) when the tap is outside the image for this row or the unit is dead
        const int oob = __builtin_amdgcn_sbfe(inval[m], tapbit[h], 1) | ((h == 1 ? !live1 : !live0) ? -1 : 0);
        ra[m][h] = buf_load4(rx, voff[m] | (unsigned)oob, (int)soff[h]);
      }
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      const unsigned so = wunit[i] ? wsoff[1] : wsoff[0];
      const unsigned off = (wbase[i] == HRSEG_BUF_OOB || so == HRSEG_BUF_OOB) ? HRSEG_BUF_OOB : wbase[i] + so;
      rwt[i] = buf_load4(rw, off, 0);
    }
    u_next += 2;
  };

  bf16x8 xf[WTM][sp_np(NS)];
  auto split_store = [&](int buf, const f32x4 (&ra)[WTM][2], const f32x4 (&rwt)[W_LOADS]) {
    unsigned char* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      u32x2 pc[sp_np(NS)];
      sp_split4<NS>(rwt[i], pc, p.wscale);
      if (tid + 256 * i < WG) {
#pragma unroll
        for (int s = 0; s < sp_np(NS); ++s) *reinterpret_cast<u32x2*>(base + s * PIECE + wst[i]) = pc[s];
      }
    }
#pragma unroll
    for (int m = 0; m < WTM; ++m) sp_split8<NS>(ra[m][0], ra[m][1], xf[m], xscale);
  };

  f32x4 acc[WTN][WTM];
#pragma unroll
  for (int n = 0; n < WTN; ++n)
#pragma unroll
    for (int m = 0; m < WTM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int foff = r16 * 64 + lds_slot(r16, g) * 16;       // weight fragment of this lane inside a 16-row tile

  // slab s lives in register set s % D; the slab count is padded to a multiple of D (dead slabs are zeros)
#pragma unroll
  for (int d = 0; d < D; ++d) issue_loads(ra[d], rwt[d]);
  split_store(0, ra[0], rwt[0]);
  __syncthreads();
  for (int s0 = 0; s0 < nslabs; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int s = s0 + d;
      issue_loads(ra[d], rwt[d]);                        // slab s + D; set d held slab s, consumed a slab ago
      const unsigned char* base = lds + (s & 1) * STAGE;
#pragma unroll
      for (int n = 0; n < WTN; ++n) {
        bf16x8 wf[sp_np(NS)];
#pragma unroll
        for (int q = 0; q < sp_np(NS); ++q) wf[q] = *reinterpret_cast<const bf16x8*>(base + q * PIECE + n * 1024 + foff);
#pragma unroll
        for (int pr = 0; pr < sp_nprod(NS); ++pr)
#pragma unroll
          for (int m = 0; m < WTM; ++m) acc[n][m] = sp_mma_p<NS>(pr, wf, xf[m], acc[n][m]);
      }
      split_store((s + 1) & 1, ra[(d + 1) % D], rwt[(d + 1) % D]);      // slab s + 1
      __syncthreads();
    }
  }

  // epilogue: lane holds channels n0+16n+4g..+3 of pixel row r16 of every tile
  // (all reads -- bias, the values an accumulating launch adds to -- before the first store: a read behind every
  // store is a memory round trip each, see igemm_patch_ws_body)
  const bool split = ks_n > 1;
  float* yrows[WTM];
  f32x4 add[WTM][WTN];
#pragma unroll
  for (int n = 0; n < WTN; ++n) {
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && ks_idx == 0) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + 16 * n + 4 * g);
#pragma unroll
    for (int m = 0; m < WTM; ++m) add[m][n] = bv;
  }
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + r16;
    yrows[m] = nullptr;
    if (row >= p.M) continue;
    size_t pix = row;
    if (!p.direct_out) {
      const int b = fdiv(row, hw, p.rcp_hw);
      const int rem = row - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      pix = (size_t)(b * p.Hy + oy * p.oys + p.oy0) * p.Wy + ox * p.oxs + p.ox0;
    }
    yrows[m] = p.y + pix * p.ldy;
    if (p.accumulate && !split) {
#pragma unroll
      for (int n = 0; n < WTN; ++n) add[m][n] += *reinterpret_cast<const f32x4*>(yrows[m] + n0 + 16 * n + 4 * g);
    }
    if (p.res && !split) {
#pragma unroll
      for (int n = 0; n < WTN; ++n) add[m][n] += *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + n0 + 16 * n + 4 * g);
    }
  }
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    float* yrow = yrows[m];
    if (!yrow) continue;
#pragma unroll
    for (int n = 0; n < WTN; ++n) {
      const int ch = n0 + 16 * n + 4 * g;
      f32x4 v = acc[n][m];
      if (NS == 4) v *= oscale;
      v += add[m][n];
      if (split) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(yrow + ch + e, v[e]);
      } else {
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(yrow + ch) = v;
      }
    }
  }
}

template <int NS, int WTM, int WTN>
__global__ __launch_bounds__(256) void igemm_sp_kernel(IgemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpLds<NS, WTN>::BYTES];
  igemm_sp_body<NS, WTM, WTN>(p, lds, blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}

// grouped form: see igemm_group_kernel
template <int NS, int WTM, int WTN, bool FULL3X3>
__global__ __launch_bounds__(256) void igemm_sp_group_kernel(IgemmGroup grp) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpLds<NS, WTN>::BYTES];
  int gi = 0;
  while (gi + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[gi]) ++gi;
  const int local = blockIdx.x - (gi ? grp.blk_end[gi - 1] : 0);
  const int tiles = grp.tiles[gi];
  igemm_sp_body<NS, WTM, WTN>(grp.a[gi], lds, local % tiles, tiles, local / tiles, grp.ksplit[gi]);
}

// --------------------------------------------------------------------------- im2col body, wide channel tiles, pre-split weights
// igemm_sp_body splits BOTH operands on the fly: per 32-wide slab and wave ~72 VALU for its pixel rows and 18 per weight
// granule, next to only 18 MFMAs on a 128 x 48 tile -- the body is VALU-bound (MFMA-busy 0.10-0.24), and a wide layer
// (720 -> 720) repeats the pixel split for each of its 15 channel tiles.  This body (fp16x2 only) takes the weights
// PRE-SPLIT from an image laid out slab by slab exactly like its LDS buffer (sp_weight_image_im2col_kernel, one small
// launch per convolution into the caller's scratch ring, as for the wave-specialised 3x3 kernels), so the weight side is
// a 16-byte copy per granule, and widens the channel tile to 16*WTN = 96 ... 240: the pixel split is amortised over up to
// 15 MFMA column tiles (90 MFMAs per slab and wave at 128 x 240) and the pixel operand is pulled through L2 3 times
// instead of 15.  Same reduction order per accumulator as igemm_sp_body (slab by slab, products in sp_mma order): results
// are bit-identical to it for ks_n == 1.
template <int WTN>
struct SpwLds {
  static constexpr int BN = 16 * WTN;
  static constexpr int WPIECE = BN * 64, WSTAGE = 2 * WPIECE;
  static constexpr int BYTES = 2 * WSTAGE;          // double buffered
};

// image of one convolution: [channel tile][slab over the full unit list][WSTAGE]
#ifdef HRSEG_TU_IM2COL
__global__ __launch_bounds__(256) void sp_weight_image_im2col_kernel(const float* __restrict__ w, unsigned char* __restrict__ img,
                                                                     int K, int T, int ntaps, unsigned long long wtap_pk,
                                                                     float wscale, int BN, int nslabs) {
  const int slab = blockIdx.x % nslabs, nt = blockIdx.x / nslabs;
  const int kch = K >> 4, nunits = ntaps * kch;
  unsigned char* dst = img + (size_t)blockIdx.x * (size_t)(2 * BN * 64);
  for (int f = threadIdx.x; f < BN * 8; f += 256) {
    const int n = f >> 3, unit = (f >> 2) & 1, gq = f & 3;
    const int u = 2 * slab + unit;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (u < nunits) {
      const int t = u / kch, c = u - t * kch;
      const int wt = (int)((wtap_pk >> (4 * t)) & 15);
      v = *reinterpret_cast<const f32x4*>(w + ((size_t)(nt * BN + n) * T + wt) * K + c * 16 + 4 * gq);
    }
    u32x2 pc[2];
    sp_split4<4>(v, pc, wscale);
    const int o = n * 64 + lds_slot(n, gq) * 16 + unit * 8;
    *reinterpret_cast<u32x2*>(dst + o) = pc[0];
    *reinterpret_cast<u32x2*>(dst + BN * 64 + o) = pc[1];
  }
}

#endif

template <int WTM, int WTN>
__device__ __forceinline__ void igemm_spw_body(const IgemmArgs& p, unsigned char* lds, const int bid, const int nblk,
                                               const int ks_idx, const int ks_n) {
  constexpr int NS = 4;
  constexpr int BM = 64 * WTM, BN = 16 * WTN;
  constexpr int WPIECE = SpwLds<WTN>::WPIECE, WSTAGE = SpwLds<WTN>::WSTAGE;
  constexpr int W16 = WSTAGE / 16;                 // 16-byte granules of a pre-split slab
  constexpr int W_LOADS = (W16 + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = p.N / BN;
  const int wg = xcd_remap(bid, nblk);
  const int nt = wg % ntn;
  const int m0 = (wg / ntn) * BM, n0 = nt * BN;

  float xscale, xinv;
  sp_pow2_scale(p.xmax, xscale, xinv);
  const float oscale = xinv * p.wscale_inv;

  const int kch = p.K >> 4;
  const int nunits = p.ntaps * kch;
  const int nslabs_all = (nunits + 1) >> 1;
  const int per = (nslabs_all + ks_n - 1) / ks_n;
  const int s_lo = ks_idx * per;
  const int s_hi = min(s_lo + per, nslabs_all);
  const int nslabs = s_hi - s_lo;

  const int hw = p.Ho * p.Wo;
  const int b0 = m0 / hw;
  const long tap0 = (long)p.oy_min * p.Wi + p.ox_min;          // <= 0
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + ((long)b0 * p.Hi * p.Wi + tap0) * p.ldx,
                                              (size_t)((long)(p.B - b0) * p.Hi * p.Wi - tap0) * p.ldx * 4);
  const __amdgpu_buffer_rsrc_t rwi = make_rsrc(reinterpret_cast<const float*>(p.wimg), (size_t)ntn * nslabs_all * WSTAGE);

  unsigned voff[WTM];
  int inval[WTM];
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + r16;
    if (row < p.M) {
      const int b = fdiv(row, hw, p.rcp_hw);
      const int rem = row - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      const int iy0 = oy * p.sy, ix0 = ox * p.sx;
      voff[m] = ((unsigned)((b - b0) * p.Hi * p.Wi + iy0 * p.Wi + ix0) * (unsigned)p.ldx + 4u * g) * 4u;
      int bad = 0;
      for (int t = 0; t < p.ntaps; ++t) {
        const int iy = iy0 + (int)((p.offy_pk >> (4 * t)) & 15) - 8, ix = ix0 + (int)((p.offx_pk >> (4 * t)) & 15) - 8;
        bad |= ((iy < 0) | (iy >= p.Hi) | (ix < 0) | (ix >= p.Wi)) ? (1 << t) : 0;
      }
      inval[m] = bad;
    } else {
      voff[m] = 0;
      inval[m] = -1;
    }
  }

  constexpr int D = SP_DEPTH;
  int u_next = 2 * s_lo;
  const int u_end = min(nunits, 2 * s_hi);
  unsigned w_off = (unsigned)(nt * nslabs_all + s_lo) * (unsigned)WSTAGE;        // image offset of the next slab to load
  const unsigned w_end = (unsigned)(nt * nslabs_all + s_hi) * (unsigned)WSTAGE;
  int u_t = __builtin_amdgcn_readfirstlane(u_next / kch), u_c = u_next - u_t * kch;      // (tap, chunk) of unit u_next, stepped (igemm_sp_body)
  f32x4 ra[D][WTM][2], rwt[D][W_LOADS];
  auto issue_loads = [&](f32x4 (&ra)[WTM][2], f32x4 (&rwt)[W_LOADS]) {
    unsigned soff[2];
    int tapbit[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int u = u_next + h;
      const bool live = u < u_end;
      const int t = live ? u_t : 0;
      const int c = live ? u_c : 0;
      if (++u_c == kch) { u_c = 0; ++u_t; }
      const int dy = (int)((p.offy_pk >> (4 * t)) & 15) - 8 - p.oy_min, dx = (int)((p.offx_pk >> (4 * t)) & 15) - 8 - p.ox_min;
      soff[h] = live ? (unsigned)((dy * p.Wi + dx) * p.ldx + 16 * c) * 4u : 0u;
      tapbit[h] = live ? t : 31;
    }
    const bool live0 = u_next < u_end, live1 = u_next + 1 < u_end;
#pragma unroll
    for (int m = 0; m < WTM; ++m) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int oob = __builtin_amdgcn_sbfe(inval[m], tapbit[h], 1) | ((h == 1 ? !live1 : !live0) ? -1 : 0);
        ra[m][h] = buf_load4(rx, voff[m] | (unsigned)oob, (int)soff[h]);
      }
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      const int f = tid + 256 * i;
      rwt[i] = buf_load4(rwi, (f < W16 && w_off < w_end) ? w_off + (unsigned)f * 16u : HRSEG_BUF_OOB, 0);
    }
    u_next += 2;
    w_off += WSTAGE;
  };

  constexpr int G = (WTN % 3 == 0) ? 3 : 2, NG = WTN / G;       // channel tiles per MFMA group, groups per slab
  constexpr int MG = G * WTM * 3;                                 // MFMAs of a group
  static_assert(WTN % G == 0, "channel tiles per block: a multiple of the group size");
  constexpr int HPG = (MG - 2 * G - 2) / 4;                       // half row splits (four steps each) a group's MFMAs carry
  static_assert(2 * NG >= W_LOADS && NG * HPG >= 2 * WTM, "a slab's MFMAs carry the whole staging of the next slab as fillers");
  unsigned xfu[2][WTM][2][4];        // pixel fragments (raw dwords) of the current slab and of the next one: [set][row tile][piece]
  f32x4 acc[WTN][WTM];
#pragma unroll
  for (int n = 0; n < WTN; ++n)
#pragma unroll
    for (int m = 0; m < WTM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int foff = r16 * 64 + lds_slot(r16, g) * 16;

  // prologue: D slabs in flight; slab s_lo staged (weights copied into buffer 0, pixel rows split into fragment set 0)
#pragma unroll
  for (int d = 0; d < D; ++d) issue_loads(ra[d], rwt[d]);
#pragma unroll
  for (int i = 0; i < W_LOADS; ++i) {
    const int f = tid + 256 * i;
    if (f < W16) *reinterpret_cast<f32x4*>(lds + f * 16) = rwt[0][i];
  }
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    bf16x8 t[2];
    sp_split8<NS>(ra[0][m][0], ra[0][m][1], t, xscale);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const u32x4 u = __builtin_bit_cast(u32x4, t[q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) xfu[0][m][q][e] = u[e];
    }
  }
  __syncthreads();
  // One slab: slab s lives in register set R = s % D and LDS buffer / fragment set C = s & 1 (the loop below is unrolled
  // 2*D times so that both are compile-time).  One wave per SIMD runs this body, so nothing hides an instruction that is
  // not issued in the shadow of an MFMA (an MFMA holds the vector issue for 8 of its 16 cycles): every piece of the next
  // slab's staging is a FILLER placed behind a particular MFMA of the current slab --
  //   MFMA 0 .. 2G-1 of group k      one ds_read_b128 each: the weight fragments of group k+1
  //   MFMA 2G, 2G+1 of group k       one 16-byte copy each of the pre-split weight slab s+1 into the other LDS buffer
  //   MFMA 2G+2 ... of group k       the four steps (scale, hi piece, residual, lo piece) of a half row split of slab s+1
  // -- pinned there by sched_barrier (machine scheduler) and by passing each piece's inputs through an empty asm (IR passes
  // would otherwise hoist the whole split to the top of the slab, where it runs with the matrix pipe idle: measured).
  // Products run product-outermost inside a group (two MFMAs on one accumulator are G*WTM apart).
  auto slab = [&](auto Rc, auto Cc) {
    constexpr int R = decltype(Rc)::value, C = decltype(Cc)::value, RN = (R + 1) % D;
    issue_loads(ra[R], rwt[R]);                          // slab s + D; set R held slab s, staged a slab ago
    const unsigned char* base = lds + C * WSTAGE;
    unsigned char* nbase = lds + (C ^ 1) * WSTAGE;
    bf16x8 wf[2][G][2];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) wf[0][j][q] = *reinterpret_cast<const bf16x8*>(base + q * WPIECE + j * 1024 + foff);
    bf16x8 xcur[WTM][2];
#pragma unroll
    for (int m = 0; m < WTM; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        xcur[m][q] = __builtin_bit_cast(bf16x8, (u32x4){xfu[C][m][q][0], xfu[C][m][q][1], xfu[C][m][q][2], xfu[C][m][q][3]});
    float hx[4];                   // the half row split in flight
    unsigned hhi[2];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int m = 0; m < WTM; ++m) {
            const int i = (pr * G + j) * WTM + m;       // MFMA index inside the group
            acc[k * G + j][m] = sp_mma_p<NS>(pr, wf[k & 1][j], xcur[m], acc[k * G + j][m]);
            if (i < 2 * G) {
              if (k + 1 < NG) {
                const int jj = i >> 1, q = i & 1;
                wf[(k + 1) & 1][jj][q] = *reinterpret_cast<const bf16x8*>(base + q * WPIECE + ((k + 1) * G + jj) * 1024 + foff);
              }
            } else if (i < 2 * G + 2) {
              const int n = 2 * k + (i - 2 * G);
              if (n < W_LOADS) {
                const int f = tid + 256 * n;
                asm volatile("" : "+v"(rwt[RN][n]));
                if (f < W16) *reinterpret_cast<f32x4*>(nbase + f * 16) = rwt[RN][n];
              }
            } else if (i < 2 * G + 2 + 4 * HPG && k * HPG + (i - (2 * G + 2)) / 4 < 2 * WTM) {
              const int h = k * HPG + (i - (2 * G + 2)) / 4;
              const int hm = h >> 1, hu = h & 1, st = (i - (2 * G + 2)) & 3;            // row tile, unit, step
              if (st == 0) {
                asm volatile("" : "+v"(ra[RN][hm][hu]));
#pragma unroll
                for (int e = 0; e < 4; ++e) hx[e] = ra[RN][hm][hu][e] * xscale;
              } else if (st == 1) {
                hhi[0] = sp_pack_f16_rtz(hx[0], hx[1]);
                hhi[1] = sp_pack_f16_rtz(hx[2], hx[3]);
              } else if (st == 2) {
                hx[0] -= sp_f16_lo(hhi[0]); hx[1] -= sp_f16_hi(hhi[0]);
                hx[2] -= sp_f16_lo(hhi[1]); hx[3] -= sp_f16_hi(hhi[1]);
              } else {
                xfu[C ^ 1][hm][0][2 * hu] = hhi[0];
                xfu[C ^ 1][hm][0][2 * hu + 1] = hhi[1];
                xfu[C ^ 1][hm][1][2 * hu] = sp_pack_f16_rne(hx[0], hx[1]);
                xfu[C ^ 1][hm][1][2 * hu + 1] = sp_pack_f16_rne(hx[2], hx[3]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
    }
    __syncthreads();
  };
  static_assert(D == 3, "unrolled by hand: 2 * D slabs per trip");
  for (int s0 = 0; s0 < nslabs; s0 += 6) {       // (slabs past the slice are zeros: every load of theirs was out of range)
    slab(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    slab(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    slab(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    slab(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    slab(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    slab(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
  }

  const bool split = ks_n > 1;
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + r16;
    if (row >= p.M) continue;
    size_t pix = row;
    if (!p.direct_out) {
      const int b = fdiv(row, hw, p.rcp_hw);
      const int rem = row - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      pix = (size_t)(b * p.Hy + oy * p.oys + p.oy0) * p.Wy + ox * p.oxs + p.ox0;
    }
    float* yrow = p.y + pix * p.ldy;
    // (the accumulators leave no registers for reading every addend ahead of the first store, as igemm_sp_body does)
#pragma unroll
    for (int n = 0; n < WTN; ++n) {
      const int ch = n0 + 16 * n + 4 * g;
      f32x4 v = acc[n][m] * oscale;
      if (p.bias && ks_idx == 0) v += *reinterpret_cast<const f32x4*>(p.bias + ch);
      if (split) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(yrow + ch + e, v[e]);
      } else {
        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(yrow + ch);
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + ch);
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(yrow + ch) = v;
      }
    }
  }
}

template <int WTM, int WTN>
__global__ __launch_bounds__(256) void igemm_spw_kernel(IgemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpwLds<WTN>::BYTES];
  igemm_spw_body<WTM, WTN>(p, lds, blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}

// --------------------------------------------------------------------------- weight gradient
// dW[co][t][ci] += sum_pix dy[pix][co] * x[pix_t][ci] on the bf16 matrix pipe.  Block = (tap, 16*TN couts,
// 16*TK cins, pixel range) as in wgrad_body; a stage is 128 pixels, 32 per wave = one K step of the MFMA.
// Both operands run over PIXELS in the reduction index, which is the strided direction of NHWC memory, so
// both tiles are staged (split on the fly) as [pixel][channel] bf16 images in LDS and read back TRANSPOSED
// with ds_read_b64_tr_b16: a 16-lane group fetches 4 pixel rows x 16 channels and every lane receives its
// channel's 4 pixels.  Two such reads make one K=32 fragment; lane group g takes pixels 4g..4g+3 and
// 16+4g..16+4g+3 of the wave's 32 (the same permutation on both operands), which keeps the two groups of
// a 32-lane half on different bank rows when the row stride is an odd multiple of 32 bytes.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 sp_tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

constexpr int sp_row_stride(int channels) {       // bytes per pixel row: 2*channels rounded up to 32 * odd
  int s = (2 * channels + 31) / 32;
  if (s % 2 == 0) ++s;
  return 32 * s;
}

template <int NS, int TN, int TK>
struct SpWgradLds {
  static constexpr int PIX = 128;
  static constexpr int SA = sp_row_stride(16 * TN), SB = sp_row_stride(16 * TK);
  static constexpr int PIECE = PIX * (SA + SB);
  static constexpr int STAGE = sp_np(NS) * PIECE;
  static constexpr int RED = 4 * TK * 256 * 4;            // cross-wave reduction, one row of tiles at a time
  static constexpr int BYTES = (STAGE > RED) ? STAGE : RED;
};

template <int NS, int TN, int TK>
__device__ __forceinline__ void wgrad_sp_body(const WgradArgs& p, unsigned char* lds, const int bx, int id) {
  using L = SpWgradLds<NS, TN, TK>;
  constexpr int PIX = L::PIX, SA = L::SA, SB = L::SB, PIECE = L::PIECE;
  constexpr int ROWS = PIX / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nct = p.Cout / (16 * TN), nkt = p.Cin / (16 * TK);
  const int kt = id % nkt;
  id /= nkt;
  const int ct = id % nct;
  const int tap = id / nct;
  const int n0 = ct * 16 * TN, k0 = kt * 16 * TK;
  const int pad = (p.ks - 1) / 2;
  const int kh = tap / p.ks - pad, kw = tap % p.ks - pad;

  const int lo = bx * p.pix_per_block;
  const int hi = min(lo + p.pix_per_block, p.M);
  const int nstages = (hi - lo + PIX - 1) / PIX;
  const int q = tid & 3, r0 = tid >> 2;
  float dyscale, dyinv;                     // fp16x2: the gradient operand is scaled by 2^14 / 2^floor(log2 |max|)
  sp_pow2_scale(p.dymax, dyscale, dyinv);

  const int hw = p.Ho * p.Wo;
  const int b_lo = lo / hw;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy + (size_t)lo * p.lddy, (size_t)max(hi - lo, 0) * p.lddy * 4);
  const __amdgpu_buffer_rsrc_t rx =
      make_rsrc(p.x + (size_t)b_lo * p.Hi * p.Wi * p.ldx, (size_t)(p.B - b_lo) * p.Hi * p.Wi * p.ldx * 4);

  f32x4 ra[ROWS][TN], rb[ROWS][TK];
  auto stage_load = [&](int s) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int ml = s * PIX + r0 + 64 * i;
      const int m = lo + ml;
      const bool ok = m < hi;
      const unsigned dyo = ok ? ((unsigned)ml * (unsigned)p.lddy + (unsigned)(n0 + 4 * q)) * 4u : HRSEG_BUF_OOB;
#pragma unroll
      for (int j = 0; j < TN; ++j) ra[i][j] = buf_load4(rdy, dyo, 64 * j);
      const int b = fdiv(m, hw, p.rcp_hw);
      const int rem = m - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      const int iy = oy * p.stride + kh, ix = ox * p.stride + kw;
      const bool okx = ok & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      const unsigned xo =
          okx ? ((unsigned)(((b - b_lo) * p.Hi + iy) * p.Wi + ix) * (unsigned)p.ldx + (unsigned)(k0 + 4 * q)) * 4u
              : HRSEG_BUF_OOB;
#pragma unroll
      for (int j = 0; j < TK; ++j) rb[i][j] = buf_load4(rx, xo, 64 * j);
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int r = r0 + 64 * i;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        u32x2 pc[sp_np(NS)];
        sp_split4<NS>(ra[i][j], pc, dyscale);
#pragma unroll
        for (int s = 0; s < sp_np(NS); ++s) *reinterpret_cast<u32x2*>(lds + s * PIECE + r * SA + (16 * j + 4 * q) * 2) = pc[s];
      }
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        u32x2 pc[sp_np(NS)];
        sp_split4<NS>(rb[i][j], pc);
#pragma unroll
        for (int s = 0; s < sp_np(NS); ++s)
          *reinterpret_cast<u32x2*>(lds + s * PIECE + PIX * SA + r * SB + (16 * j + 4 * q) * 2) = pc[s];
      }
    }
  };

  f32x4 acc[TN][TK];
#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int k = 0; k < TK; ++k) acc[n][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read addresses: lane 16g+i supplies row (i>>2) of its group's 4-pixel block, columns 4(i&3)..+3
  const int g = lane >> 4, li = lane & 15;
  const int prow = wave * 32 + 4 * g + (li >> 2);
  const int aoff = prow * SA + (li & 3) * 8, boff = PIX * SA + prow * SB + (li & 3) * 8;

  if (nstages > 0) {
    stage_load(0);
    stage_store();
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const bool more = s + 1 < nstages;
    if (more) stage_load(s + 1);
    bf16x8 bfr[TK][sp_np(NS)];
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
      for (int pc = 0; pc < sp_np(NS); ++pc) {
        const s16x4 v0 = sp_tr_read(lds + pc * PIECE + boff + k * 32);
        const s16x4 v1 = sp_tr_read(lds + pc * PIECE + boff + k * 32 + 16 * SB);
        bfr[k][pc] = __builtin_bit_cast(bf16x8, (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
      }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      bf16x8 afr[sp_np(NS)];
#pragma unroll
      for (int pc = 0; pc < sp_np(NS); ++pc) {
        const s16x4 v0 = sp_tr_read(lds + pc * PIECE + aoff + n * 32);
        const s16x4 v1 = sp_tr_read(lds + pc * PIECE + aoff + n * 32 + 16 * SA);
        afr[pc] = __builtin_bit_cast(bf16x8, (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
      }
#pragma unroll
      for (int pr = 0; pr < sp_nprod(NS); ++pr)
#pragma unroll
        for (int k = 0; k < TK; ++k) acc[n][k] = sp_mma_p<NS>(pr, afr, bfr[k], acc[n][k]);
    }
    __syncthreads();                       // every wave is done reading before the image is rewritten
    if (more) stage_store();
    __syncthreads();
  }

  // cross-wave reduction, one row of tiles at a time: red[wave][k][reg*64 + lane] (fp32)
  float* red = reinterpret_cast<float*>(lds);
  const int r = tid >> 6, l = tid & 63;
#pragma unroll
  for (int n = 0; n < TN; ++n) {
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wave * TK + k) * 256 + e * 64 + lane] = acc[n][k][e];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) v += red[(wv * TK + k) * 256 + tid];
      const int co = n0 + 16 * n + 4 * (l >> 4) + r;  // D row = 4*(lane>>4)+reg
      const int ci = k0 + 16 * k + (l & 15);          // D col = lane&15
      atomicAdd(p.dw + ((size_t)co * p.T + tap) * p.Cin + ci, NS == 4 ? v * dyinv : v);
    }
    __syncthreads();
  }
}

template <int NS, int TN, int TK>
__global__ __launch_bounds__(256) void wgrad_sp_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpWgradLds<NS, TN, TK>::BYTES];
  const int tiles = gridDim.y, nblk = gridDim.x * gridDim.y;
  const int r = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, nblk);
  wgrad_sp_body<NS, TN, TK>(p, lds, r / tiles, r % tiles);
}

// --------------------------------------------------------------------------- wide-tile weight gradient
// The tap-per-block body above gives every wave its own 32 pixels of a stage and the WHOLE tile, so the tile is bounded by
// one wave's accumulators (80 x 80) and every operand row is pulled Cin / 80 resp. Cout / 80 times: on the 720 -> 720 layer
// 10 GB through L2 at the ~7.5 TB/s that path delivers = the 1.32 ms the launch takes (MFMA-busy 0.18).  Here the NWR x NWC
// waves of a block share the 32 pixels of a stage and each owns a (16 TN) x (16 TK) part of the block's (16 TN NWR) x (16 TK NWC)
// tile (240 x 144 with 3 x 3 waves of 80 x 48): per pixel 384 operand channels are pulled for 34,560 outputs instead of 160
// for 6,400, no cross-wave reduction, one atomic add per element and block.  Single LDS buffer, the next stage prefetched
// into registers across the MFMAs; 576 threads, 60 KB of LDS, one block per CU.
template <int NS, int TN, int TK, int NWR, int NWC>
struct SpWgradWideLds {
  static constexpr int PIX = 32;
  static constexpr int CA = 16 * TN * NWR, CB = 16 * TK * NWC;        // block tile: output channels x input channels
  static constexpr int SA = sp_row_stride(CA), SB = sp_row_stride(CB);
  static constexpr int PIECE = PIX * (SA + SB);
  static constexpr int BYTES = sp_np(NS) * PIECE;
};

template <int NS, int TN, int TK, int NWR, int NWC>
__device__ __forceinline__ void wgrad_spw_body(const WgradArgs& p, unsigned char* lds, const int bx, int id) {
  using L = SpWgradWideLds<NS, TN, TK, NWR, NWC>;
  constexpr int PIX = L::PIX, SA = L::SA, SB = L::SB, PIECE = L::PIECE, CA = L::CA, CB = L::CB;
  constexpr int NT = 64 * NWR * NWC;
  constexpr int QA = CA / 4, QB = CB / 4;                              // float4 columns of a pixel row
  constexpr int LA = (PIX * QA + NT - 1) / NT, LB = (PIX * QB + NT - 1) / NT;      // loads per thread and stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / NWC, wc = wave % NWC;
  const int nct = p.Cout / CA, nkt = p.Cin / CB;
  const int kt = id % nkt;
  id /= nkt;
  const int ct = id % nct;
  const int tap = id / nct;
  const int n0 = ct * CA, k0 = kt * CB;
  const int pad = (p.ks - 1) / 2;
  const int kh = tap / p.ks - pad, kw = tap % p.ks - pad;
  const int lo = bx * p.pix_per_block;
  const int hi = min(lo + p.pix_per_block, p.M);
  const int nstages = (hi - lo + PIX - 1) / PIX;
  float dyscale, dyinv;
  sp_pow2_scale(p.dymax, dyscale, dyinv);
  const int hw = p.Ho * p.Wo;
  const int b_lo = lo / hw;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy + (size_t)lo * p.lddy, (size_t)max(hi - lo, 0) * p.lddy * 4);
  const __amdgpu_buffer_rsrc_t rx =
      make_rsrc(p.x + (size_t)b_lo * p.Hi * p.Wi * p.ldx, (size_t)(p.B - b_lo) * p.Hi * p.Wi * p.ldx * 4);

  f32x4 ra[LA], rb[LB];
  auto stage_load = [&](int s) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int u = tid + NT * i, r = u / QA, cq = u - r * QA;
      const int ml = s * PIX + r;
      const bool ok = (u < PIX * QA) & (lo + ml < hi);
      ra[i] = buf_load4(rdy, ok ? ((unsigned)ml * (unsigned)p.lddy + (unsigned)(n0 + 4 * cq)) * 4u : HRSEG_BUF_OOB, 0);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int u = tid + NT * i, r = u / QB, cq = u - r * QB;
      const int m = lo + s * PIX + r;
      const int b = fdiv(m, hw, p.rcp_hw);
      const int rem = m - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      const int iy = oy * p.stride + kh, ix = ox * p.stride + kw;
      const bool ok = (u < PIX * QB) & (m < hi) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      rb[i] = buf_load4(rx, ok ? ((unsigned)(((b - b_lo) * p.Hi + iy) * p.Wi + ix) * (unsigned)p.ldx + (unsigned)(k0 + 4 * cq)) * 4u
                               : HRSEG_BUF_OOB, 0);
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int u = tid + NT * i, r = u / QA, cq = u - r * QA;
      u32x2 pc[sp_np(NS)];
      sp_split4<NS>(ra[i], pc, dyscale);
      if (u < PIX * QA)
#pragma unroll
        for (int q = 0; q < sp_np(NS); ++q) *reinterpret_cast<u32x2*>(lds + q * PIECE + r * SA + cq * 8) = pc[q];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int u = tid + NT * i, r = u / QB, cq = u - r * QB;
      u32x2 pc[sp_np(NS)];
      sp_split4<NS>(rb[i], pc);
      if (u < PIX * QB)
#pragma unroll
        for (int q = 0; q < sp_np(NS); ++q) *reinterpret_cast<u32x2*>(lds + q * PIECE + PIX * SA + r * SB + cq * 8) = pc[q];
    }
  };

  f32x4 acc[TN][TK];
#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int k = 0; k < TK; ++k) acc[n][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read addresses (as in wgrad_sp_body; all waves read the SAME 32 pixels, their own channel columns)
  const int g = lane >> 4, li = lane & 15;
  const int prow = 4 * g + (li >> 2);
  const int aoff = prow * SA + (li & 3) * 8 + wr * TN * 32, boff = PIX * SA + prow * SB + (li & 3) * 8 + wc * TK * 32;

  if (nstages > 0) {
    stage_load(0);
    stage_store();
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const bool more = s + 1 < nstages;
    if (more) stage_load(s + 1);
    bf16x8 afr[TN][sp_np(NS)];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int pc = 0; pc < sp_np(NS); ++pc) {
        const s16x4 v0 = sp_tr_read(lds + pc * PIECE + aoff + n * 32);
        const s16x4 v1 = sp_tr_read(lds + pc * PIECE + aoff + n * 32 + 16 * SA);
        afr[n][pc] = __builtin_bit_cast(bf16x8, (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
      }
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      bf16x8 bfr[sp_np(NS)];
#pragma unroll
      for (int pc = 0; pc < sp_np(NS); ++pc) {
        const s16x4 v0 = sp_tr_read(lds + pc * PIECE + boff + k * 32);
        const s16x4 v1 = sp_tr_read(lds + pc * PIECE + boff + k * 32 + 16 * SB);
        bfr[pc] = __builtin_bit_cast(bf16x8, (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
      }
#pragma unroll
      for (int pr = 0; pr < sp_nprod(NS); ++pr)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[n][k] = sp_mma_p<NS>(pr, afr[n], bfr, acc[n][k]);
    }
    __syncthreads();                       // every wave is done reading before the image is rewritten
    if (more) stage_store();
    __syncthreads();
  }

#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = n0 + (wr * TN + n) * 16 + 4 * g + e;       // D row = 4*(lane>>4)+reg
        const int ci = k0 + (wc * TK + k) * 16 + li;              // D col = lane&15
        atomicAdd(p.dw + ((size_t)co * p.T + tap) * p.Cin + ci, NS == 4 ? acc[n][k][e] * dyinv : acc[n][k][e]);
      }
}

template <int NS, int TN, int TK, int NWR, int NWC>
__global__ __launch_bounds__(64 * NWR * NWC) void wgrad_spw_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpWgradWideLds<NS, TN, TK, NWR, NWC>::BYTES];
  const int tiles = gridDim.y, nblk = gridDim.x * gridDim.y;
  const int r = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, nblk);
  wgrad_spw_body<NS, TN, TK, NWR, NWC>(p, lds, r / tiles, r % tiles);
}

// several problems in one launch (the fuse layers' 1x1 / stride-2 weight gradients of an HRNet module): blocks
// [blk_end[g-1], blk_end[g]) belong to problem g, each with its own pixel ranges x (tap, tile) blocks
template <int NS, int TN, int TK>
__global__ __launch_bounds__(256) void wgrad_sp_group_kernel(WgradGroup grp) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpWgradLds<NS, TN, TK>::BYTES];
  int g = 0;
  while (g + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[g]) ++g;
  const int lo = g ? grp.blk_end[g - 1] : 0;
  const int nblk = grp.blk_end[g] - lo;
  const int tiles = nblk / grp.gx[g];
  const int r = xcd_remap(blockIdx.x - lo, nblk);
  wgrad_sp_body<NS, TN, TK>(grp.a[g], lds, r / tiles, r % tiles);
}

// --------------------------------------------------------------------------- 3x3 stride-1: halo patch in LDS
// At bf16 matrix rates the im2col body above is bound by the L2 -> CU path: every input pixel is pulled nine
// times (once per tap) for only 16*WTN output channels (measured: ~8 TB/s of operand traffic whatever the
// number of products).  For 3x3 stride-1 convolutions (forward and data gradient) this body stages the
// (TH+2) x 18 input patch of a TH x 16 output tile ONCE per K stage, already split into bf16 pieces (one split
// per element instead of nine), and the nine taps read their pixel fragments from LDS at shifted positions.
//   patch image : [piece][16-channel chunk][patch pixel][32 B]; the four 8-byte granules of a row are XOR-ed
//                 with 2*((pixel>>3)&1): 16 consecutive pixels x 2 granules = one conflict-free ds_read_b64
//   reduction   : units (tap, chunk) of the K stage, flattened; a slab = two consecutive units (so 48-channel
//                 stages pair chunk 2 of one tap with chunk 0 of the next); weight slabs stream through a
//                 double-buffered LDS image as in igemm_sp_body
//   wave w owns tile rows [w*RPW, (w+1)*RPW), all 16 columns, all 16*WTN channels.
template <int NS, int TH, int WTN, int CS>
struct SpPatchLds {
  static constexpr int PP = (TH + 2) * 18;              // patch pixels
  // bytes per chunk image, padded to 64 mod 256: a 16-lane store group holds one pixel's granules of all CS chunks,
  // and chunk images a multiple of 128 bytes apart would put chunks 0 and 2 on the same banks
  static constexpr int CHUNK = PP * 32 + ((64 - (PP * 32) % 256) + 256) % 256;
  static constexpr int PPIECE = CS * CHUNK;             // per piece
  static constexpr int PATCH = sp_np(NS) * PPIECE;
  static constexpr int WPIECE = 16 * WTN * 64, WSTAGE = sp_np(NS) * WPIECE;
  static constexpr int BYTES = PATCH + 3 * WSTAGE;      // weight slabs: three buffers (fragments are read one slab ahead)
};

// Work list of a block: output tiles first, first + stride, ... (< end), each with K / (16*CS) K stages.  The
// slab loop of a K stage is fully unrolled (tap offsets, chunk indices and register-set parity are compile-time
// constants; FLIP selects the data-gradient tap geometry), so a slab costs its LDS reads, its MFMAs, one
// weight-slab load + split + store and ONE barrier.  Software pipeline per slab:
//   top    : weight + pixel FRAGMENTS of slab s+1 are read from LDS into the second register set; global
//            loads of the weight slab s+2 are issued (and, two slabs before a K stage ends, of the next
//            patch: next K stage of this tile or first K stage of the block's next tile)
//   middle : the MFMAs of slab s on the register set filled during slab s-1 -- they wait for nothing
//   bottom : weight slab s+2 is split and stored into the third LDS buffer; at a K-stage boundary the next
//            patch replaces the current one (its last fragments were read a slab, i.e. a barrier, earlier)
template <int NS, int TH, int WTN, int CS, int FLIP>
__device__ __forceinline__ void igemm_patch_sp_body(const IgemmArgs& p, unsigned char* lds, const int first, const int stride,
                                                    const int end) {
  using L = SpPatchLds<NS, TH, WTN, CS>;
  constexpr int RPW = TH / 4, BN = 16 * WTN, PW = 18, PP = L::PP;
  constexpr int PG = PP * CS * 4;                          // 16-byte granules of one K stage of the patch
  constexpr int P_LOADS = (PG + 255) / 256;
  constexpr int WG = BN * 8, W_LOADS = (WG + 255) / 256;
  constexpr int NU = 9 * CS, NSLAB = (NU + 1) / 2;         // units / slabs per K stage
  static_assert(NSLAB % 2 == 0, "register-set parity must restart with every K stage");
  static_assert(WG >= 256 && PG >= 256, "the spare lanes of a partial round repeat a granule of the round before");
  unsigned char* lpatch = lds;
  unsigned char* lw = lds + L::PATCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int H = p.Ho, W = p.Wo;                            // stride 1, pad 1: input and output are H x W
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + TH - 1) / TH;
  const int ntn = p.N / BN;
  const int nks = p.K / (16 * CS);
  if (first >= end) return;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (size_t)p.N * p.T * p.K * 4);
  float xscale, xinv;                       // fp16x2: power-of-two scale of the pixel operand (a gradient: from its |max|)
  sp_pow2_scale(p.xmax, xscale, xinv);
  const float oscale = xinv * p.wscale_inv;

  struct Geom { int b, y0, x0, n0; };
  auto tile_geom = [&](int t) {       // channel tile fastest, then tile column, tile row, image
    Geom q;
    const int nt = t % ntn;
    int mt = t / ntn;
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    q.b = mt / tiles_y;
    q.y0 = ty * TH; q.x0 = tx * 16; q.n0 = nt * BN;
    return q;
  };

  // patch granules of this thread: f = tid + 256 i -> (patch pixel, chunk, 4-channel group)
  f32x4 rp[P_LOADS];
  auto patch_load = [&](const Geom& q, int ks) {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)q.b * H * W * p.ldx, (size_t)H * W * p.ldx * 4);
#pragma unroll
    for (int i = 0; i < P_LOADS; ++i) {
      int f = tid + 256 * i;
      if (f >= PG) f -= 256;          // the spare lanes of the last round repeat a granule of the round before (no branch)
      const int pix = f / (CS * 4), rem = f - pix * (CS * 4);
      const int py = pix / PW, px = pix - py * PW;
      const int iy = q.y0 - 1 + py, ix = q.x0 - 1 + px;
      const bool ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
      const unsigned off = ok ? (unsigned)(iy * W + ix) * (unsigned)(p.ldx * 4) + (unsigned)(rem * 16) : HRSEG_BUF_OOB;
      rp[i] = buf_load4(rx, off, ks * CS * 64);
    }
  };
  auto patch_store = [&]() {
#pragma unroll
    for (int i = 0; i < P_LOADS; ++i) {
      u32x2 pc[sp_np(NS)];
      sp_split4<NS>(rp[i], pc, xscale);
      int f = tid + 256 * i;
      if (f >= PG) f -= 256;
      const int pix = f / (CS * 4), rem = f - pix * (CS * 4);
      const int c = rem >> 2, q = rem & 3;
      const int o = c * L::CHUNK + pix * 32 + ((q ^ (2 * ((pix >> 3) & 1))) << 3);
#pragma unroll
      for (int s = 0; s < sp_np(NS); ++s) *reinterpret_cast<u32x2*>(lpatch + s * L::PPIECE + o) = pc[s];
    }
  };

  // weight granules: f -> (row n = f>>3, unit (f>>2)&1, 4-channel group f&3), as in igemm_sp_body
  unsigned wrow[W_LOADS];
  int wst[W_LOADS];
#pragma unroll
  for (int i = 0; i < W_LOADS; ++i) {
    int f = tid + 256 * i;
    if (f >= WG) f -= 256;            // spare lanes repeat a granule of the round before: same data to the same slot
    const int n = f >> 3, unit = (f >> 2) & 1, gq = f & 3;
    wrow[i] = ((unsigned)n * (unsigned)(p.T * p.K) + 4u * gq) * 4u;
    wst[i] = n * 64 + lds_slot(n, gq) * 16 + unit * 8;
  }
  const bool wunit1 = (tid >> 2) & 1;        // (f>>2)&1 is the same for every i (256 is a multiple of 8)
  f32x4 rwt[2][W_LOADS];        // two slabs in flight: loaded a slab before they are split and stored
  // p.wimg (fp16x2, set by the host when it has scratch space): the weights come pre-split, slab by slab in the
  // layout of the LDS buffer (sp_weight_image_kernel), and are only copied -- the split of a weight slab is two
  // thirds of this body's vector arithmetic, repeated for every tile
  constexpr int W16 = NS == 4 ? L::WSTAGE / 16 : 0;        // (images exist for the two-piece fp16 split only)
  static_assert(W16 <= 256 * W_LOADS, "a pre-split slab fits the register set of an fp32 one");
  const bool img = NS == 4 && p.wimg != nullptr;
  const __amdgpu_buffer_rsrc_t rwi = make_rsrc(reinterpret_cast<const float*>(p.wimg), (size_t)ntn * nks * NSLAB * L::WSTAGE);
  // weight slab `slab` (compile-time after unrolling) of K stage ks for channel tile n0 -> register set `set`
  auto w_load = [&](int n0, int ks, int slab, int set) {
    if (img) {
      const unsigned base = (unsigned)(((n0 / BN) * nks + ks) * NSLAB + slab) * (unsigned)L::WSTAGE;
#pragma unroll
      for (int i = 0; i < W_LOADS; ++i) {
        const int f = tid + 256 * i;
        rwt[set][i] = buf_load4(rwi, f < W16 ? base + (unsigned)f * 16u : HRSEG_BUF_OOB, 0);
      }
      return;
    }
    const int uA = 2 * slab, uB = 2 * slab + 1;
    const int tA = uA / CS, cA = uA - tA * CS, tB = uB / CS, cB = uB - tB * CS;      // weight tap index = tap
    const unsigned col0 = (unsigned)(n0 * p.T * p.K + ks * CS * 16) * 4u;
    const unsigned sA = col0 + (unsigned)(tA * p.K + cA * 16) * 4u;
    const unsigned sB = (uB < NU) ? col0 + (unsigned)(tB * p.K + cB * 16) * 4u : HRSEG_BUF_OOB;
    const unsigned so = wunit1 ? sB : sA;
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      const unsigned off = (so == HRSEG_BUF_OOB) ? HRSEG_BUF_OOB : wrow[i] + so;
      rwt[set][i] = buf_load4(rw, off, 0);
    }
  };
  auto w_store = [&](int wboff, int set) {
    unsigned char* base = lw + wboff;
    if (img) {
#pragma unroll
      for (int i = 0; i < W_LOADS; ++i) {
        const int f = tid + 256 * i;
        if (f < W16) *reinterpret_cast<f32x4*>(base + f * 16) = rwt[set][i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      u32x2 pc[sp_np(NS)];
      sp_split4<NS>(rwt[set][i], pc, p.wscale);
#pragma unroll
      for (int s = 0; s < sp_np(NS); ++s) *reinterpret_cast<u32x2*>(base + s * L::WPIECE + wst[i]) = pc[s];
    }
  };

  const int foff = r16 * 64 + lds_slot(r16, g) * 16;       // weight fragment inside a 16-row tile
  const int prow0 = (wave * RPW) * PW + r16;               // patch pixel of (tile row wave*RPW, column r16) at tap offset (0,0)
  // this lane's pixel-fragment byte offsets for patch pixels prow0 + d: the granule swizzle depends on bit 3 of
  // the pixel index, so keep both variants of (pixel*32 + granule*8) and pick per (compile-time) offset d
  const int pbase = prow0 * 32;

  // fragments of slab `slab` (compile-time) from the current patch and weight buffer offset wboff
  auto read_frags = [&](int slab, int wboff, bf16x8 (&xf)[RPW][sp_np(NS)], bf16x8 (&wf)[WTN][sp_np(NS)]) {
    const int uA = 2 * slab, uB = 2 * slab + 1;
    const int tA = uA / CS, cA = uA - tA * CS;
    const int tB = (uB < NU) ? uB / CS : 0, cB = (uB < NU) ? uB - tB * CS : 0;
    const int dA = (FLIP ? 2 - tA / 3 : tA / 3) * PW + (FLIP ? 2 - tA % 3 : tA % 3);
    const int dB = (FLIP ? 2 - tB / 3 : tB / 3) * PW + (FLIP ? 2 - tB % 3 : tB % 3);
#pragma unroll
    for (int m = 0; m < RPW; ++m) {
      const int pa = prow0 + m * PW + dA, pb = prow0 + m * PW + dB;
      const int oa = cA * L::CHUNK + pbase + (m * PW + dA) * 32 + ((g ^ (2 * ((pa >> 3) & 1))) << 3);
      const int ob = cB * L::CHUNK + pbase + (m * PW + dB) * 32 + ((g ^ (2 * ((pb >> 3) & 1))) << 3);
#pragma unroll
      for (int q = 0; q < sp_np(NS); ++q) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(lpatch + q * L::PPIECE + oa);
        u32x2 hi = u32x2{0u, 0u};
        if (uB < NU) hi = *reinterpret_cast<const u32x2*>(lpatch + q * L::PPIECE + ob);
        xf[m][q] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
      }
    }
    const unsigned char* base = lw + wboff;
#pragma unroll
    for (int n = 0; n < WTN; ++n)
#pragma unroll
      for (int q = 0; q < sp_np(NS); ++q) wf[n][q] = *reinterpret_cast<const bf16x8*>(base + q * L::WPIECE + n * 1024 + foff);
  };

  f32x4 acc[WTN][RPW];
  auto zero_acc = [&]() {
#pragma unroll
    for (int n = 0; n < WTN; ++n)
#pragma unroll
      for (int m = 0; m < RPW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_tile = [&](const Geom& q) {
    // all reads of the tile before its first store (see igemm_patch_ws_body)
    f32x4 add[RPW][WTN];
#pragma unroll
    for (int n = 0; n < WTN; ++n) {
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + q.n0 + 16 * n + 4 * g);
#pragma unroll
      for (int m = 0; m < RPW; ++m) add[m][n] = bv;
    }
    if (p.accumulate) {
#pragma unroll
      for (int m = 0; m < RPW; ++m) {
        const int oy = q.y0 + wave * RPW + m, ox = q.x0 + r16;
        if (oy >= H || ox >= W) continue;
        const float* yrow = p.y + ((size_t)(q.b * H + oy) * W + ox) * p.ldy;
#pragma unroll
        for (int n = 0; n < WTN; ++n) add[m][n] += *reinterpret_cast<const f32x4*>(yrow + q.n0 + 16 * n + 4 * g);
      }
    }
    if (p.res) {
#pragma unroll
      for (int m = 0; m < RPW; ++m) {
        const int oy = q.y0 + wave * RPW + m, ox = q.x0 + r16;
        if (oy >= H || ox >= W) continue;
        const float* rrow = p.res + ((size_t)(q.b * H + oy) * W + ox) * p.ldr;
#pragma unroll
        for (int n = 0; n < WTN; ++n) add[m][n] += *reinterpret_cast<const f32x4*>(rrow + q.n0 + 16 * n + 4 * g);
      }
    }
#pragma unroll
    for (int m = 0; m < RPW; ++m) {
      const int oy = q.y0 + wave * RPW + m, ox = q.x0 + r16;
      if (oy >= H || ox >= W) continue;
      float* yrow = p.y + ((size_t)(q.b * H + oy) * W + ox) * p.ldy;
#pragma unroll
      for (int n = 0; n < WTN; ++n) {
        f32x4 v = acc[n][m];
        if (NS == 4) v *= oscale;
        v += add[m][n];
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(yrow + q.n0 + 16 * n + 4 * g) = v;
      }
    }
  };

  bf16x8 xfr[2][RPW][sp_np(NS)], wfr[2][WTN][sp_np(NS)];
  int t = first;
  Geom cur = tile_geom(t);
  // prologue: patch of the first K stage, weight slabs 0 and 1, fragments of slab 0
  patch_load(cur, 0);
  w_load(cur.n0, 0, 0, 0);
  w_load(cur.n0, 0, 1, 1);
  patch_store();
  w_store(0, 0);
  w_store(L::WSTAGE, 1);
  w_load(cur.n0, 0, 2, 0);           // stored by the first slab of the loop
  __syncthreads();
  read_frags(0, 0, xfr[0], wfr[0]);
  zero_acc();
  int wb = 0;                        // byte offset of the CURRENT slab's weight buffer (three buffers, rotating)
  for (;;) {
    bool have_next = false;
    for (int ks = 0; ks < nks; ++ks) {
      const bool last_ks = ks + 1 == nks;
      const int nt = last_ks ? t + stride : t;
      const int nks_ = last_ks ? 0 : ks + 1;
      have_next = nt < end;
      Geom nxt = cur;
      if (last_ks && have_next) nxt = tile_geom(nt);
#pragma unroll
      for (int s = 0; s < NSLAB; ++s) {
        const int wb1 = (wb == 2 * L::WSTAGE) ? 0 : wb + L::WSTAGE;
        const int wb2 = (wb1 == 2 * L::WSTAGE) ? 0 : wb1 + L::WSTAGE;
        // weight slab s+3 -> registers (split + stored a slab later); s+2 is in the other register set
        if (s + 3 < NSLAB) w_load(cur.n0, ks, s + 3, (s + 1) & 1);
        else if (have_next) w_load(nxt.n0, nks_, s + 3 - NSLAB, (s + 1) & 1);
        if (s == NSLAB - 3 && have_next) patch_load(nxt, nks_);          // in flight behind three slabs of MFMAs
        if (s + 1 < NSLAB) read_frags(s + 1, wb1, xfr[(s + 1) & 1], wfr[(s + 1) & 1]);
#pragma unroll
        for (int pr = 0; pr < sp_nprod(NS); ++pr)
#pragma unroll
          for (int n = 0; n < WTN; ++n)
#pragma unroll
            for (int m = 0; m < RPW; ++m) acc[n][m] = sp_mma_p<NS>(pr, wfr[s & 1][n], xfr[s & 1][m], acc[n][m]);
        if (s + 2 < NSLAB || have_next) w_store(wb2, s & 1);
        if (s == NSLAB - 1) {
          if (last_ks) { store_tile(cur); zero_acc(); }
          if (have_next) patch_store();
        }
        __syncthreads();
        if (s == NSLAB - 1 && have_next) read_frags(0, wb1, xfr[0], wfr[0]);  // first slab of the new patch
        wb = wb1;
      }
      if (last_ks) cur = nxt;
    }
    if (!have_next) break;
    t += stride;
  }
}

// two blocks (2 waves per SIMD) per CU where the instance fits: the bf16x3 instances (three pieces of every fragment and of
// the staging registers) and the 96 x 64 fp16x2 / bf16x2 tilings need more than 256 registers for that and run one block per CU
constexpr int sp_patch_min_waves(int ns, int wtn, int cs) { return (ns == 3 || (wtn == 6 && cs == 4)) ? 1 : 2; }
template <int NS, int TH, int WTN, int CS, int FLIP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sp_patch_min_waves(NS, WTN, CS), 2)))
void igemm_patch_sp_kernel(IgemmArgs p, int ntotal) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpPatchLds<NS, TH, WTN, CS>::BYTES];
  // a block owns `chunk` consecutive tiles (neighbours in x share halo columns: L2 hits in time)
  const int chunk = (ntotal + gridDim.x - 1) / gridDim.x;
  const int first = blockIdx.x * chunk;
  igemm_patch_sp_body<NS, TH, WTN, CS, FLIP>(p, lds, first, 1, min(first + chunk, ntotal));
}

// --------------------------------------------------------------------------- halo patch, wave-specialised (8 waves)
// In igemm_patch_sp_body every wave does everything: its MFMA burst is followed by the split + store of the next
// weight slab, the fragment reads and the barrier, and two waves per SIMD from two blocks do not hide that (MFMA-busy
// 0.30-0.34).  Here a 512-thread block splits the roles.  Waves 0-3 (one per SIMD) are CONSUMERS: per slab they issue
// the MFMAs of the current slab in an order that never puts two dependent products back to back, with the LDS reads
// of the NEXT slab's fragments pinned between them (sched_barrier keeps the compiler from sinking the reads to their
// uses), and nothing else.  Waves 4-7 are PRODUCERS.  The roles share their SIMD's vector issue: an MFMA leaves 8 of its 16
// cycles free, and every vector / LDS / memory instruction of EITHER wave beyond that is paid in matrix-pipe time (measured,
// DESIGN.md section 3) -- the split of roles moves the staging instructions, it does not make them free, so the producer is
// written for instruction count: the weights come PRE-SPLIT from a global image laid out slab by slab exactly like the LDS
// buffer (splitting a weight slab on the fly is ~30 instructions per thread and slab; the image is persistent, rebuilt once
// per model call) and the producers only copy them -- global -> registers two slabs ahead -> ds_write_b128 two slabs ahead of
// their first read -- and stage the next K stage's patch into the second patch buffer from per-block granule tables.  One
// barrier per slab orders both roles.  One block per CU, persistent over a range of tiles.
template <int NS, int TH, int WTN, int CS>
struct SpPatchWsLds {
  using P = SpPatchLds<NS, TH, WTN, CS>;
  static constexpr int STAT_MAXN = 1024;                           // BatchNorm statistics in the epilogue: [2][N] fp64 sums per block
  static constexpr int STAT = 2 * STAT_MAXN * 8;
  static constexpr int BYTES = 2 * P::PATCH + 3 * P::WSTAGE + STAT;       // two patch buffers, three weight slabs, the sums
  static constexpr int NSLAB = (9 * CS + 1) / 2;
};

// weight image: for every (channel tile, K stage, slab) the WSTAGE bytes the LDS weight buffer holds for it; one
// launch writes the images of all problems of a grouped launch
template <int NS, int WTN, int CS>
__device__ __forceinline__ void sp_weight_image_body(const float* __restrict__ w, unsigned char* __restrict__ img, int K,
                                                     float wscale, int blk) {
  using L = SpPatchLds<NS, 8, WTN, CS>;
  constexpr int BN = 16 * WTN, WG = BN * 8, NU = 9 * CS, NSLAB = (NU + 1) / 2;
  const int nks = K / (16 * CS);
  const int slab = blk % NSLAB;
  const int ks = (blk / NSLAB) % nks;
  const int nt = blk / (NSLAB * nks);
  unsigned char* dst = img + (size_t)blk * L::WSTAGE;
  for (int f = threadIdx.x; f < WG; f += 256) {
    const int n = f >> 3, unit = (f >> 2) & 1, gq = f & 3;
    const int u = 2 * slab + unit;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (u < NU) {
      const int t = u / CS, c = u - t * CS;
      v = *reinterpret_cast<const f32x4*>(w + ((size_t)(nt * BN + n) * 9 + t) * K + ks * CS * 16 + c * 16 + 4 * gq);
    }
    u32x2 pc[sp_np(NS)];
    sp_split4<NS>(v, pc, wscale);
    const int o = n * 64 + lds_slot(n, gq) * 16 + unit * 8;
#pragma unroll
    for (int q = 0; q < sp_np(NS); ++q) *reinterpret_cast<u32x2*>(dst + q * L::WPIECE + o) = pc[q];
  }
}
struct WeightImageGroup {
  int n;
  int ns;                  // 4: two fp16 pieces per weight (fp16x2), 1: one bf16 piece (bf16)
  int blk_end[MAXG];
  int kind[MAXG];          // channel tiling of the wave-specialised body (conv.hip: ws_kind)
  int K[MAXG];
  float wscale[MAXG];
  const float* w[MAXG];
  unsigned char* img[MAXG];
};
// table form: the images of EVERY registered convolution weight in one launch (hrseg_weight_images_refresh); the table lives
// in device memory, a block finds its entry by bisection over the running block counts
struct WeightImageTabEntry {
  const float* w;
  unsigned char* img;
  int K, kind, ns, blk_end;
  float wscale;
  int pad;
};
#ifdef HRSEG_TU_WS          // non-template kernels are defined in the one translation unit that launches them
__global__ __launch_bounds__(256) void sp_weight_image_table_kernel(const WeightImageTabEntry* __restrict__ tab, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= tab[mid].blk_end) lo = mid + 1; else hi = mid;
  }
  const WeightImageTabEntry e = tab[lo];
  const int blk = blockIdx.x - (lo ? tab[lo - 1].blk_end : 0);
  if (e.ns == 1) {
    if (e.kind == 2) sp_weight_image_body<1, 6, 3>(e.w, e.img, e.K, e.wscale, blk);
    else if (e.kind == 3) sp_weight_image_body<1, 4, 4>(e.w, e.img, e.K, e.wscale, blk);
    else sp_weight_image_body<1, 3, 3>(e.w, e.img, e.K, e.wscale, blk);
    return;
  }
  if (e.kind == 2) sp_weight_image_body<4, 6, 3>(e.w, e.img, e.K, e.wscale, blk);
  else if (e.kind == 3) sp_weight_image_body<4, 4, 4>(e.w, e.img, e.K, e.wscale, blk);
  else sp_weight_image_body<4, 3, 3>(e.w, e.img, e.K, e.wscale, blk);
}
__global__ __launch_bounds__(256) void sp_weight_image_kernel(WeightImageGroup g) {
  int gi = 0;
  while (gi + 1 < g.n && (int)blockIdx.x >= g.blk_end[gi]) ++gi;
  const int blk = blockIdx.x - (gi ? g.blk_end[gi - 1] : 0);
  const int kind = g.kind[gi];
  if (g.ns == 1) {
    if (kind == 2) sp_weight_image_body<1, 6, 3>(g.w[gi], g.img[gi], g.K[gi], g.wscale[gi], blk);
    else if (kind == 3) sp_weight_image_body<1, 4, 4>(g.w[gi], g.img[gi], g.K[gi], g.wscale[gi], blk);
    else sp_weight_image_body<1, 3, 3>(g.w[gi], g.img[gi], g.K[gi], g.wscale[gi], blk);
    return;
  }
  if (kind == 2) sp_weight_image_body<4, 6, 3>(g.w[gi], g.img[gi], g.K[gi], g.wscale[gi], blk);
  else if (kind == 3) sp_weight_image_body<4, 4, 4>(g.w[gi], g.img[gi], g.K[gi], g.wscale[gi], blk);
  else sp_weight_image_body<4, 3, 3>(g.w[gi], g.img[gi], g.K[gi], g.wscale[gi], blk);
}
#endif

// sum over the 16 lanes of a DPP row, left in lane 15 of the row (row_shr:1,2,4,8, zero fill for lanes shifted in)
__device__ __forceinline__ float sp_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}

// s_waitcnt vmcnt(n) tied to the register a load fills (n: a constant once the slab loop is unrolled)
__device__ __forceinline__ void sp_wait_vm(f32x4& r, int n) {
  switch (n) {
#define HRSEG_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(r)); break;
    HRSEG_VM(0) HRSEG_VM(1) HRSEG_VM(2) HRSEG_VM(3) HRSEG_VM(4) HRSEG_VM(5) HRSEG_VM(6) HRSEG_VM(7) HRSEG_VM(8) HRSEG_VM(9)
    HRSEG_VM(10) HRSEG_VM(11) HRSEG_VM(12) HRSEG_VM(13) HRSEG_VM(14) HRSEG_VM(15) HRSEG_VM(16) HRSEG_VM(17) HRSEG_VM(18) HRSEG_VM(19)
    HRSEG_VM(20) HRSEG_VM(21) HRSEG_VM(22) HRSEG_VM(23) HRSEG_VM(24) HRSEG_VM(25) HRSEG_VM(26) HRSEG_VM(27) HRSEG_VM(28) HRSEG_VM(29)
    HRSEG_VM(30) HRSEG_VM(31) HRSEG_VM(32) HRSEG_VM(33) HRSEG_VM(34) HRSEG_VM(35) HRSEG_VM(36) HRSEG_VM(37) HRSEG_VM(38) HRSEG_VM(39)
    HRSEG_VM(40) HRSEG_VM(41) HRSEG_VM(42) HRSEG_VM(43) HRSEG_VM(44) HRSEG_VM(45) HRSEG_VM(46) HRSEG_VM(47) HRSEG_VM(48)
#undef HRSEG_VM
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(r)); break;
  }
}

template <int NS, int TH, int WTN, int CS, int FLIP>
__device__ __forceinline__ void igemm_patch_ws_body(const IgemmArgs& p, unsigned char* lds, const int first, const int end,
                                                    const int block_row = 0) {
  static_assert(NS == 4 || NS == 1, "the wave-specialised body: fp16x2 (two pieces, three products) or bf16 (one piece, one product)");
  using L = SpPatchLds<NS, TH, WTN, CS>;
  constexpr int RPW = TH / 4, BN = 16 * WTN, PW = 18, PP = L::PP;
  // patch granules (16 bytes of fp32 = 4 channels): a round of the 256 producer threads covers PR whole pixels,
  // thread -> (pixel ptid / GPP within the round, granule ptid % GPP of the pixel), so that a granule's pixel is
  // pix0 + PR * i with no division in the loop (CS = 3: 252 threads work, 4 idle)
  constexpr int GPP = CS * 4, PR = 256 / GPP;
  constexpr int P_LOADS = (PP + PR - 1) / PR;              // rounds per K stage
  constexpr int W16 = L::WSTAGE / 16, W_LOADS = (W16 + 255) / 256;     // 16-byte granules of a pre-split weight slab
  constexpr int NU = 9 * CS, NSLAB = (NU + 1) / 2;
#ifndef HRSEG_WS_EXP
#define HRSEG_WS_EXP 0       // MEASUREMENT ONLY (wrong results): 1 no MFMAs, 2 no weight loads, 4 no patch loads, 8 no weight stores, 16 no epilogue, 32 epilogue stores out of range
#endif
  static_assert(NSLAB % 2 == 0, "register-set parity must restart with every K stage");
  unsigned char* lpatch = lds;                             // [2][PATCH]
  unsigned char* lw = lds + 2 * L::PATCH;                  // [3][WSTAGE]
  // BatchNorm statistics of the OUTPUT in the epilogue (p.stat_partial, training forward): the block keeps [2][N] fp64 sums
  // (sum y, sum y^2 per output channel over every pixel of its tiles) in LDS and writes them as ONE row of the partial-sum
  // buffer the BatchNorm finalize kernel reads -- the statistics kernel, its launch boundary and its pass over y (measured:
  // 2.6 ms of a 51 ms step with every statistics launch left out) disappear for the layers this kernel produces.
  double* lstat = reinterpret_cast<double*>(lds + 2 * L::PATCH + 3 * L::WSTAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool consumer = __builtin_amdgcn_readfirstlane(wave) < 4;      // wave-uniform: a scalar branch
  const int ptid = tid & 255;
  const int r16 = lane & 15, g = lane >> 4;
  const int H = p.Ho, W = p.Wo;
  // CANVAS mode (p.cv_w1 > 0, chosen by the host for images whose 16-column tiles are mostly padding): the cv_nb images of
  // the batch lie side by side on one canvas, a zero column between neighbours (the 3x3 halo of one image never sees the
  // next), and the tiles cover the canvas: 39 x 39 images 1.26x -> 1.05x padded area, 20 x 20 images 1.92x -> 1.32x.  A canvas
  // column cx belongs to image cx / w1 (exact multiply-high for cx, w1 < 2^16: host-checked), column cx % w1, the gap column
  // w1 - 1 = W being invalid.  Plain mode is the same arithmetic with magic 0 (image 0 of a descriptor that starts at the
  // tile's own image): one code path, results identical pixel for pixel (same K stages, slabs and product order).
  const int cv_w1 = p.cv_w1, cv_nb = p.cv_w1 > 0 ? p.cv_nb : 1;
  const unsigned cv_magic = p.cv_w1 > 0 ? p.cv_magic : 0u;
  const int Wc = p.cv_w1 > 0 ? cv_nb * cv_w1 - 1 : W;          // canvas width in pixels
  const int tiles_x = (Wc + 15) >> 4, tiles_y = (H + TH - 1) / TH;
  const int ntn = p.N / BN;
  const int nks = p.K / (16 * CS);
  if (first >= end) return;
#ifndef HRSEG_WS_STAMP
#define HRSEG_WS_STAMP 0     // MEASUREMENT ONLY: block 0's waves 0 (consumer) and 4 (producer) write the time they ARRIVE at every slab barrier
#endif                       // and the time they LEAVE it into p.stat_partial ([role][slab][2] 64-bit counters; no statistics then)
#if HRSEG_WS_STAMP
  const bool stat = false;
  unsigned long long* const stamp_out = reinterpret_cast<unsigned long long*>(p.stat_partial);
  const bool stamping = stamp_out != nullptr && block_row == 0 && (wave == 0 || wave == 4);
  int stamp_j = 0;
  auto stamped_barrier = [&]() {
    unsigned long long t0, t1;
    asm volatile("s_memtime %0" : "=s"(t0));
    __builtin_amdgcn_s_barrier();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (stamping && stamp_j < 1024 && lane == 0) {
      unsigned long long* o = stamp_out + ((wave >> 2) * 1024 + stamp_j) * 2;
      o[0] = t0;
      o[1] = t1;
    }
    ++stamp_j;
  };
#else
  const bool stat = p.stat_partial != nullptr;
  auto stamped_barrier = [&]() { __builtin_amdgcn_s_barrier(); };
#endif
  if (stat)
    for (int i = tid; i < 2 * p.N; i += 512) lstat[i] = 0.0;            // (ordered before the first epilogue by the prologue barrier)

  // tile t -> (channel tile fastest, then tile column, tile row, image); walked incrementally (the divisions
  // happen once per block: a producer wave has no issue slots to spare for them)
  struct Geom { int b, y0, x0, nt; };
  auto tile_geom = [&](int t) {
    Geom q;
    q.nt = t % ntn;
    int mt = t / ntn;
    const int tx = mt % tiles_x;
    mt /= tiles_x;
    const int ty = mt % tiles_y;
    q.b = mt / tiles_y;
    q.y0 = ty * TH; q.x0 = tx * 16;
    return q;
  };
  auto tile_next = [&](Geom& q) {
    if (++q.nt < ntn) return;
    q.nt = 0;
    q.x0 += 16;
    if (q.x0 < tiles_x * 16) return;
    q.x0 = 0;
    q.y0 += TH;
    if (q.y0 < tiles_y * TH) return;
    q.y0 = 0;
    ++q.b;
  };

  if (consumer) {
    float xscale, xinv;
    sp_pow2_scale(p.xmax, xscale, xinv);
    const float oscale = xinv * p.wscale_inv;
    const int foff = r16 * 64 + lds_slot(r16, g) * 16;
    const int prow0 = (wave * RPW) * PW + r16;
    const int pbase = prow0 * 32;
    constexpr int NP = sp_np(NS);
    constexpr int UNITS = RPW * NP + WTN * NP;             // fragment registers (8 halfs each) of a slab
    constexpr int MM = WTN * RPW * sp_nprod(NS);           // MFMAs of a slab
    // fragment r of slab `slab` (compile-time) -> register set
    // A pixel fragment's LDS address is (patch pixel prow0 + c) * 32 + the 8-byte slot g, swizzled by bit 3 of the pixel index,
    // c = m * PW + tap offset a compile-time constant.  The swizzle depends on c only through c mod 16, so SIXTEEN per-lane base
    // registers (xbase[j]: the lane's pixel prow0, slot swizzled for c = j mod 16, the patch buffer the reads currently target)
    // serve every fragment read with an immediate offset -- left to itself the compiler keeps one address register per
    // (row, tap) pair, ~40-57 loop-invariant registers on the 16-row tiling, which is what made these kernels spill.
    int xbase[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xbase[j] = pbase + ((g ^ (2 * (((prow0 + j) >> 3) & 1))) << 3);
    auto xbase_flip = [&](int to_buf) {
#pragma unroll
      for (int j = 0; j < 16; ++j) xbase[j] += to_buf ? L::PATCH : -L::PATCH;
    };
    auto read_unit = [&](int slab, int wboff, int r, bf16x8 (&xf)[RPW][NP], bf16x8 (&wf)[WTN][NP]) {
      if (r < RPW * NP) {
        const int m = r / NP, q = r % NP;
        const int uA = 2 * slab, uB = 2 * slab + 1;
        const int tA = uA / CS, cA = uA - tA * CS;
        const int tB = (uB < NU) ? uB / CS : 0, cB = (uB < NU) ? uB - tB * CS : 0;
        const int dA = (FLIP ? 2 - tA / 3 : tA / 3) * PW + (FLIP ? 2 - tA % 3 : tA % 3);
        const int dB = (FLIP ? 2 - tB / 3 : tB / 3) * PW + (FLIP ? 2 - tB % 3 : tB % 3);
        const unsigned char* pb = lpatch + q * L::PPIECE;
        const int ca = m * PW + dA, cb = m * PW + dB;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(pb + xbase[ca & 15] + (cA * L::CHUNK + ca * 32));
        u32x2 hi = u32x2{0u, 0u};
        if (uB < NU) hi = *reinterpret_cast<const u32x2*>(pb + xbase[cb & 15] + (cB * L::CHUNK + cb * 32));
        xf[m][q] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
      } else {
        const int i = r - RPW * NP, n = i / NP, q = i % NP;
        wf[n][q] = *reinterpret_cast<const bf16x8*>(lw + wboff + q * L::WPIECE + n * 1024 + foff);
      }
    };
    f32x4 acc[WTN][RPW];
    auto zero_acc = [&]() {
#pragma unroll
      for (int n = 0; n < WTN; ++n)
#pragma unroll
        for (int m = 0; m < RPW; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // Epilogue in two parts.  fetch_add: everything the tile's output ADDS to its accumulators -- bias, the values an
    // accumulating launch (data gradient into an existing gradient) adds to, the fused residual -- read into registers;
    // store_acc: scale, add, ReLU, store.  Every read is issued before the first store (for all the compiler knows a store
    // may alias the next read, and a read behind every store is a memory round trip each: 156 -> 140 us on the accumulating
    // data-gradient launches).  Tilings with registers to spare inside the grouped kernel's allocation (PF: 48- and 64-channel
    // tiles on 8 rows, 158 / 178 of the 246 registers the 96-channel tiling makes the kernel allocate anyway) issue fetch_add
    // S_PF slabs BEFORE the tile's last slab: a 48-channel layer has ONE K stage of 14 slabs (3.5 us) per tile, and reading
    // the accumulate values at the very end exposed a memory round trip per tile (+23 us per accumulating launch).  Same
    // arithmetic, same order: bit-identical results.
    constexpr bool PF = WTN * RPW <= 8;
    constexpr int S_PF = NSLAB > 10 ? NSLAB - 10 : 0;
    f32x4 add[RPW][WTN];
    // The epilogue's launch parameters, pinned in scalar registers: `p` lives in the kernel-argument segment, and left alone
    // the compiler re-loads its fields where they are used -- one s_load + s_waitcnt lgkmcnt(0) round trip per output row of
    // EVERY tile's epilogue, with the matrix pipe idle (a tile cost ~2 us beyond its slabs; a 48-channel layer's tile is 4.6 us
    // of slabs).  A value that went through v_readfirstlane cannot be rematerialised from memory.
    auto pin_i = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto pin_p = [](const void* ptr) {
      const unsigned long long a = (unsigned long long)ptr;
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
      return (unsigned long long)lo | ((unsigned long long)hi << 32);
    };
    float* const e_y = reinterpret_cast<float*>(pin_p(p.y));
    const float* const e_bias = reinterpret_cast<const float*>(pin_p(p.bias));
    const float* const e_res = reinterpret_cast<const float*>(pin_p(p.res));
    const int e_ldy = pin_i(p.ldy), e_ldr = pin_i(p.ldr), e_acc = pin_i(p.accumulate), e_relu = pin_i(p.relu);
    const int e_early = pin_i(p.epi_early), e_stat = pin_i(stat ? 1 : 0), e_N = pin_i(p.N);
    constexpr bool SREG = PF;                                  // per-lane statistics sums kept in registers over the block's tiles
    const int e_sreg = pin_i((stat && ntn == 1) ? 1 : 0);
    f32x4 rs1[SREG ? WTN : 1], rs2[SREG ? WTN : 1];
#pragma unroll
    for (int n = 0; n < (SREG ? WTN : 1); ++n) rs1[n] = rs2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Output addressing: buffer descriptors based at the tile's first image (canvas mode: the cv_nb images of the canvas), 32-bit
    // byte offsets per lane, HRSEG_BUF_OOB for a pixel outside the image -- its load returns zeros, its store writes nothing: no
    // exec-mask branch per output row and no 64-bit address arithmetic in an epilogue that runs with the matrix pipe idle.
    // (Host-checked: cv_nb * H * W * ld * 4 bytes < 2^31 for y and the residual, conv.hip ws_kind / ws_canvas.)
    const unsigned ldy4 = (unsigned)e_ldy * 4u, ldr4 = (unsigned)e_ldr * 4u;
    const int wave_rows = pin_i(wave * RPW);               // first tile row of this wave
    const unsigned g16 = (unsigned)g * 16u;                // this lane's four channels inside a 16-channel tile, bytes
    auto tile_offsets = [&](const Geom& q, unsigned ld4, unsigned (&off)[RPW]) {
      // this lane's output column: canvas column -> (image, column); invalid on the gap column and past the last image.
      // One offset per tile (two 32-bit multiplies, quarter rate), the rows of the tile a scalar stride apart.
      const int cxo = q.x0 + r16;
      const int obc = (int)__umulhi((unsigned)cxo, cv_magic);
      const int ox = cxo - obc * cv_w1;
      const bool ook = !(HRSEG_WS_EXP & 32) & (ox < W) & (obc < cv_nb);
      const int oy0 = q.y0 + wave_rows;
      const unsigned base = (unsigned)((obc * H + oy0) * W + ox) * ld4 + (unsigned)(q.nt * BN) * 4u + g16;
      const unsigned rowstep = (unsigned)W * ld4;
#pragma unroll
      for (int m = 0; m < RPW; ++m) off[m] = (ook & (oy0 + m < H)) ? base + (unsigned)m * rowstep : HRSEG_BUF_OOB;
    };
    const int e_nothing = pin_i((!p.bias && !p.accumulate && !p.res) ? 1 : 0);
    auto fetch_add = [&](const Geom& q) {
      if (e_nothing) {             // the training forward and the plain data gradient: zeros, behind ONE scalar branch (the general
#pragma unroll                    // path below spent ~700 cycles per tile on its three not-taken cases: tools/ws_stamps.py)
        for (int m = 0; m < RPW; ++m)
#pragma unroll
          for (int n = 0; n < WTN; ++n) add[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
      }
#pragma unroll
      for (int n = 0; n < WTN; ++n) {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e_bias) bv = *reinterpret_cast<const f32x4*>(e_bias + q.nt * BN + 16 * n + 4 * g);
#pragma unroll
        for (int m = 0; m < RPW; ++m) add[m][n] = bv;
      }
      if (e_acc) {
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(e_y + (size_t)q.b * H * W * e_ldy, (size_t)cv_nb * H * W * ldy4);
        unsigned off[RPW];
        tile_offsets(q, ldy4, off);
#pragma unroll
        for (int m = 0; m < RPW; ++m)
#pragma unroll
          for (int n = 0; n < WTN; ++n) add[m][n] += buf_load4(ry, off[m], 64 * n);
      }
      if (e_res) {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(e_res + (size_t)q.b * H * W * e_ldr, (size_t)cv_nb * H * W * ldr4);
        unsigned off[RPW];
        tile_offsets(q, ldr4, off);
#pragma unroll
        for (int m = 0; m < RPW; ++m)
#pragma unroll
          for (int n = 0; n < WTN; ++n) add[m][n] += buf_load4(rr, off[m], 64 * n);
      }
    };
    auto store_acc_as = [&](const Geom& q, auto plain_tag) {
      constexpr bool PLAIN = decltype(plain_tag)::value;
      const __amdgpu_buffer_rsrc_t ry = make_rsrc(e_y + (size_t)q.b * H * W * e_ldy, (size_t)cv_nb * H * W * ldy4);
      unsigned off[RPW];
      tile_offsets(q, ldy4, off);
      if (e_stat) {
        // per channel tile: this lane's sums over its RPW rows (fp32, RPW terms).  Pixels outside the image contribute nothing.
        // SREG (one channel tile per pixel tile and registers to spare: the 48- and 64-channel layers, whose tiles are the
        // shortest): the lane keeps adding into its own fp32 sums over ALL tiles of the block (a block walks a few dozen tiles)
        // and the cross-lane reduction happens once, at the end of the block (flush_stats).  Otherwise per tile: a 16-lane
        // row reduction over the tile's 16 columns (row_shr 1, 2, 4, 8 with zero fill: lane 15 of the row ends with the
        // total, 32 fp32 terms), then ONE fp64 LDS atomic per channel and sum from that lane.
#pragma unroll
        for (int n = 0; n < WTN; ++n) {
          f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < RPW; ++m) {
            const bool ok = off[m] != HRSEG_BUF_OOB;
            const f32x4 v = PLAIN ? acc[n][m] * oscale : acc[n][m] * oscale + add[m][n];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ve = ok ? v[e] : 0.f;
              s1[e] += ve;
              s2[e] += ve * ve;
            }
          }
          if (SREG && e_sreg) {
            rs1[n] += s1;
            rs2[n] += s2;
            continue;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s1[e] = sp_row16_sum(s1[e]);
            s2[e] = sp_row16_sum(s2[e]);
          }
          if (r16 == 15) {
            const int c0 = q.nt * BN + 16 * n + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              __hip_atomic_fetch_add(lstat + c0 + e, (double)s1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_add(lstat + e_N + c0 + e, (double)s2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < RPW; ++m) {
#pragma unroll
        for (int n = 0; n < WTN; ++n) {
          f32x4 v = PLAIN ? acc[n][m] * oscale : acc[n][m] * oscale + add[m][n];
          if (!PLAIN && e_relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          buf_store4(ry, off[m], 64 * n, v);
        }
        // A 16-byte store reads its data registers over several cycles after it issues, and the compiler lets a packed
        // multiply of the NEXT row write them in the very next slot (measured on the plain path: the last dword of a row's
        // last store came out overwritten in the last four lanes of each row of 16, in 0.5 % of the tiles, run to run different).
        // Nothing crosses this point, and the next vector write is eight wait states away.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // (A specialisation for the training forward -- nothing to add, no ReLU: multiply in place and store -- measured no faster,
    // and its in-place multiplies are what exposed the store-data hazard above: one code path.)
    auto store_acc = [&](const Geom& q) { store_acc_as(q, std::false_type{}); };
    bf16x8 xfr[2][RPW][NP], wfr[2][WTN][NP];
    Geom cur = tile_geom(first);
    __syncthreads();                       // the producers' prologue: first patch, weight slabs 0 and 1
#pragma unroll
    for (int r = 0; r < UNITS; ++r) read_unit(0, 0, r, xfr[0], wfr[0]);
    zero_acc();
    int wb = 0, pb = 0;                    // weight buffer offset of the CURRENT slab, patch buffer of the CURRENT stage
    for (int t = first;; ++t) {
      bool have_next = false;
      for (int ks = 0; ks < nks; ++ks) {
        const bool last_ks = ks + 1 == nks;
        have_next = (last_ks ? t + 1 : t) < end;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
          const int wb1 = (wb == 2 * L::WSTAGE) ? 0 : wb + L::WSTAGE;
          // the next slab's fragments: slab s+1 of this patch, or slab 0 of the next stage's patch (complete since
          // the barrier before this slab; past the block's last stage the reads fetch stale data nobody uses)
          const int nslab = (s + 1 < NSLAB) ? s + 1 : 0;
          if (s + 1 == NSLAB) xbase_flip(pb ^ 1);          // (every read of a stage's last slab targets the next stage's patch)
          int k = 0;
#pragma unroll
          for (int pr = 0; pr < sp_nprod(NS); ++pr) {
#pragma unroll
            for (int n = 0; n < WTN; ++n) {
#pragma unroll
              for (int m = 0; m < RPW; ++m) {
                // products in the order of sp_mma (w1 x0, w0 x1, w0 x0) per accumulator, WTN*RPW MFMAs apart
                if (NS == 4) {
                  const f16x8 wv = __builtin_bit_cast(f16x8, wfr[s & 1][n][pr == 0 ? NP - 1 : 0]);
                  const f16x8 xv = __builtin_bit_cast(f16x8, xfr[s & 1][m][pr == 1 ? NP - 1 : 0]);
#if HRSEG_WS_EXP & 1
                  asm volatile("" :: "v"(wv), "v"(xv));
#else
                  acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, acc[n][m], 0, 0, 0);
#endif
                } else {
                  acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[s & 1][n][0], xfr[s & 1][m][0], acc[n][m], 0, 0, 0);
                }
                // the next slab's fragment reads, spread over this slab's MFMAs (front-loading them so that the last MFMAs cover
                // their latency measured no different: the wait before the slab barrier is not where the slab's time goes)
#pragma unroll
                for (int r = k * UNITS / MM; r < (k + 1) * UNITS / MM; ++r)
                  read_unit(nslab, wb1, r, xfr[(s + 1) & 1], wfr[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                ++k;
              }
            }
          }
#if HRSEG_WS_EXP & 16
          if (s == NSLAB - 1 && last_ks) {
            if (acc[0][0][0] == 1.2345f) store_acc(cur);
            zero_acc();
          }
#else
          if (PF && s == S_PF && last_ks && e_early) fetch_add(cur);
          if (s == NSLAB - 1 && last_ks) {
#if HRSEG_WS_STAMP
            unsigned long long e0, e1, e2, e3;                // (measurement build: the epilogue's phases of wave 0, block 0)
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e0) :: "memory");
            if (!(PF && e_early)) fetch_add(cur);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e1) :: "memory");
            store_acc(cur);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e2) :: "memory");
            zero_acc();
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e3) :: "memory");
            if (stamping && wave == 0 && lane == 0 && stamp_j < 1024) {
              unsigned long long* o = stamp_out + 4096 + stamp_j * 4;
              o[0] = e0; o[1] = e1; o[2] = e2; o[3] = e3;
            }
#else
            if (!(PF && e_early)) fetch_add(cur);
            store_acc(cur);
            zero_acc();
#endif
          }
#endif
          // The slab barrier orders LDS traffic only (the producers' ds_writes against these reads).  __syncthreads() would
          // also wait for every outstanding vector-memory operation of this wave (its workgroup-scope fence emits vmcnt(0)):
          // the tile's stores after store_acc -- a memory round trip per tile before the next tile's first slab -- and the
          // loads of fetch_add issued S_PF slabs early.  Nothing another wave of the block reads depends on them, so the
          // consumers wait for their LDS operations and meet the barrier directly.
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          stamped_barrier();
          wb = wb1;
        }
        pb ^= 1;
        if (last_ks) tile_next(cur);
      }
      if (!have_next) break;
    }
    if (SREG && e_sreg) {          // the register sums of this wave: row reduction, then one fp64 LDS atomic per channel and sum
#pragma unroll
      for (int n = 0; n < WTN; ++n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = sp_row16_sum(rs1[n][e]), b = sp_row16_sum(rs2[n][e]);
          if (r16 == 15) {
            __hip_atomic_fetch_add(lstat + 16 * n + 4 * g + e, (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(lstat + e_N + 16 * n + 4 * g + e, (double)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
  } else {
    // PRODUCERS.  A wave issues an instruction every four or five cycles at best and the block meets at ONE barrier per slab, so a
    // slab lasts as long as its slowest wave's instruction stream: a 36-MFMA slab is 576 cycles of the matrix pipe, i.e. ~120
    // producer instructions.  (Round 4 measured the previous flat-loop producer -- ~230 instructions per slab on the 16-row
    // tiling, a third of them per-granule address arithmetic with quarter-rate 32-bit multiplies -- as what set the slab time:
    // 2,240 / 1,340-1,710 cycles per slab on 16-row / 96-channel tiles, 1,450 / 990-1,140 with the loads and their address
    // arithmetic compiled out, tools/ws_bound.py.)  This producer runs K stage by K stage with the slab loop of a stage unrolled
    // (every condition on the slab index folds away), and everything about a patch granule that does not depend on the tile --
    // its pixel of the patch, its offset from the patch origin, its LDS address -- is computed ONCE per block:
    //   * weights: slab j+2 goes from the register ring (loaded D slabs earlier) to LDS, slab j+2+D is loaded into the same
    //     registers: one address add per load / store group, the image offset of the stage is a scalar;
    //   * patch of the NEXT stage, granule by granule (loaded at slab s, split and stored at slab s+D): an interior tile's
    //     granule loads with its precomputed offset as the vector offset and the tile origin as the scalar offset -- no
    //     vector instruction at all; a border or canvas tile pays eight (row / column range tests against per-stage scalars);
    //   * waits are counted: the number of loads younger than the one a store needs is a compile-time function of the slab.
    float xscale, xinv;
    sp_pow2_scale(p.xmax, xscale, xinv);
    const int per_tile = nks * NSLAB;                      // slabs (= weight image entries) per tile
    auto rsrc_words = [](const void* base, size_t bytes) {
      const unsigned long long a = (unsigned long long)base;
      i32x4_t r;
      r[0] = (int)(unsigned)(a & 0xffffffffull);
      r[1] = (int)(unsigned)((a >> 32) & 0xffffull);
      r[2] = (int)(unsigned)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes);
      r[3] = HRSEG_BUF_FLAGS;
      return r;
    };
    auto ld16 = [](f32x4& dst, const i32x4_t& rs, unsigned voff, unsigned soff) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    const i32x4_t rw = rsrc_words(p.wimg, (size_t)ntn * per_tile * L::WSTAGE);
    // ---- what a thread's patch granules are, once per block
    const int pix0 = ptid / GPP, prem = ptid - pix0 * GPP;
    const bool pwork = pix0 < PR;                          // (CS = 3: the last four threads stage no patch granule)
    const int pst0 = (prem >> 2) * L::CHUNK, pq = prem & 3;
    const unsigned ldx4 = (unsigned)p.ldx * 4u;
    unsigned rel[P_LOADS];          // byte offset of the granule from the patch origin (row y0-1, column x0-1) of a plain image
    int lst[P_LOADS];               // its LDS byte offset inside a patch buffer (piece 0)
    int gpy[P_LOADS], gpx[P_LOADS]; // its patch row and column (row: a value no image reaches for a granule that does not exist)
#pragma unroll
    for (int i = 0; i < P_LOADS; ++i) {
      const int pix = pix0 + PR * i;
      const int py = (pix * 3641) >> 16, px = pix - py * PW;           // pix / 18 for pix < 2^12
      const bool ex = pwork & (pix < PP);
      rel[i] = ex ? (unsigned)(py * W + px) * ldx4 + (unsigned)prem * 16u : HRSEG_BUF_OOB;
      // a granule that does not exist stores zeros into the padding behind chunk 0's pixels (CHUNK is PP * 32 + 192 bytes)
      lst[i] = ex ? pst0 + pix * 32 + ((pq ^ (2 * ((pix >> 3) & 1))) << 3) : PP * 32 + (lane & 15) * 8;
      gpy[i] = ex ? py : 0x40000000;
      gpx[i] = px;
    }
    static_assert(L::CHUNK - PP * 32 >= 128 + 8, "the padding of a chunk image takes the stores of granules that do not exist");
    // weights: 16-byte granule f = ptid + 256 i of a slab
    unsigned wv[W_LOADS];
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) wv[i] = (ptid + 256 * i < W16) ? (unsigned)(ptid + 256 * i) * 16u : 0x80000000u;
    const bool wlast = __builtin_amdgcn_readfirstlane((int)((wave - 4) * 64 + 256 * (W_LOADS - 1) < W16)) != 0;   // wave-uniform
    const int wl0 = (int)(lw - lds) + ptid * 16;
    const int exp_nosplit = p.exp_nosplit | p.x_presplit;        // x stored pre-split: copy, do not split (see hrseg.h x_split)
    // ---- per-stage scalars of the stage whose patch is being staged
    struct Stage { unsigned tb, tb0, delta; int y0m1, x0m1, xlo, xspan, xb; bool fast, second; };
    const unsigned hw = (unsigned)(H * W);
    auto stage_scalars = [&](const Geom& q) {
      Stage z;
      const int x0m1 = q.x0 - 1;
      z.y0m1 = q.y0 - 1;
      z.x0m1 = x0m1;
      z.xlo = x0m1 < 0 ? 1 : 0;
      if (cv_w1 > 0) {
        const int bc0 = x0m1 < 0 ? 0 : (int)__umulhi((unsigned)x0m1, cv_magic);
        z.xb = (bc0 + 1) * cv_w1 - x0m1;
        z.second = bc0 + 1 < cv_nb;
        z.tb0 = ((unsigned)((bc0 * H + z.y0m1) * W + x0m1 - bc0 * cv_w1)) * ldx4;
        z.delta = (hw - (unsigned)cv_w1) * ldx4;
        z.fast = false;
        if (bc0 >= cv_nb) z.xb = z.xlo + 1;               // (a tile column past the last image: nothing valid)
      } else {
        z.xb = W - x0m1 + 1;
        z.second = false;
        z.tb0 = (unsigned)(z.y0m1 * W + x0m1) * ldx4;
        z.delta = 0u;
        z.fast = (q.y0 >= 1) & (q.y0 + TH + 1 <= H) & (q.x0 >= 1) & (q.x0 + 17 <= W);
      }
      z.xspan = z.xb - 1 - z.xlo;
      z.tb = z.tb0;
      return z;
    };
    const bool cv_narrow = cv_w1 > 0 && cv_w1 < 18;       // an 18-column patch may span three images: exact per-granule arithmetic
    auto patch_voff = [&](const Stage& z, int i) -> unsigned {          // border / canvas tile: the granule's offset or OOB
      if (cv_narrow) {
        const int iy = gpy[i] + z.y0m1, cx = gpx[i] + z.x0m1;
        const int bc = (int)__umulhi((unsigned)cx, cv_magic);
        const int ix = cx - bc * cv_w1;
        const bool ok = ((unsigned)iy < (unsigned)H) & (cx >= 0) & (ix < W) & (bc < cv_nb);
        return ok ? (unsigned)((bc * H + iy) * W + ix) * ldx4 + (unsigned)prem * 16u : HRSEG_BUF_OOB;
      }
      const bool oky = (unsigned)(gpy[i] + z.y0m1) < (unsigned)H;
      const bool ok1 = (unsigned)(gpx[i] - z.xlo) < (unsigned)z.xspan;
      const bool ok2 = z.second & (gpx[i] >= z.xb);
      const unsigned off = rel[i] + z.tb0 + (ok2 ? z.delta : 0u);
      return (oky & (ok1 | ok2)) ? off : HRSEG_BUF_OOB;
    };
    auto patch_store1 = [&](const f32x4& v, int i, int pboff) {
      u32x2 pc[sp_np(NS)];
      if (exp_nosplit) {
        const u32x4 raw = __builtin_bit_cast(u32x4, v);
#pragma unroll
        for (int s = 0; s < sp_np(NS); ++s) pc[s] = u32x2{raw[(2 * s) & 3], raw[(2 * s + 1) & 3]};
      } else {
        sp_split4<NS>(v, pc, xscale);
      }
      const int o = pboff + lst[i];
#pragma unroll
      for (int s = 0; s < sp_np(NS); ++s) *reinterpret_cast<u32x2*>(lpatch + s * L::PPIECE + o) = pc[s];
    };
    auto image_rsrc = [&](int b) { return rsrc_words(p.x + (size_t)b * H * W * p.ldx, (size_t)cv_nb * H * W * p.ldx * 4); };
    // image offset of the weight slabs of stage (channel tile nt, K stage ks)
    auto stage_woff = [&](int nt, int ks) { return (unsigned)((nt * nks + ks) * NSLAB) * (unsigned)L::WSTAGE; };
    // Registers in flight.  Weights: a ring of DW = 2 slab sets -- a stage has an even number of slabs, so every stage starts at
    // ring position 0 and the loop's back edge maps each register to itself.  (The loads are inline assembly the compiler knows
    // nothing about: a ring whose phase alternates between stages needs two code instances and a join between them, and at
    // that join the compiler MOVED ring registers whose loads were still in flight -- measured, wrong results.)  Patch: one
    // register set per granule of the stage, loaded GPL per slab from slab 0 and stored DP slabs later -- nothing of it is in
    // flight across a stage boundary.
    constexpr int DW = 2;
    constexpr int GPL = (P_LOADS + NSLAB - 6) / (NSLAB - 5);           // granules loaded per slab so that DP >= 4
    constexpr int DP = NSLAB - 1 - (P_LOADS + GPL - 1) / GPL;          // the last granule is stored at slab NSLAB - 2
    static_assert(DP >= 4 && (P_LOADS + GPL - 1) / GPL <= NSLAB - DW, "patch schedule");
    f32x4 rw4[DW][W_LOADS], pg[P_LOADS];
    Geom cur = tile_geom(first);
    // ---- prologue: the first patch and weight slabs 0, 1 into LDS, slabs 2 .. D+1 in flight
    unsigned woff_cur = stage_woff(cur.nt, 0);             // weight image offset of the CURRENT stage's slab 0
    {
      f32x4 v[P_LOADS];
      const i32x4_t rx0 = image_rsrc(cur.b);
      const Stage z0 = stage_scalars(cur);
#pragma unroll
      for (int i = 0; i < P_LOADS; ++i) ld16(v[i], rx0, patch_voff(z0, i), 0u);
      f32x4 w01[2][W_LOADS];
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i) ld16(w01[d][i], rw, wv[i] + woff_cur + (unsigned)(d * L::WSTAGE), 0u);
#pragma unroll
      for (int i = 0; i < P_LOADS; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i]));
#pragma unroll
      for (int i = 0; i < P_LOADS; ++i) patch_store1(v[i], i, 0);
#pragma unroll
      for (int i = 0; i < W_LOADS; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w01[0][i]), "+v"(w01[1][i]));
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i)
          if (i + 1 < W_LOADS || wlast) *reinterpret_cast<f32x4*>(lds + wl0 + d * L::WSTAGE + i * 4096) = w01[d][i];
    }
    // the stage after the current one (its patch is staged during the current stage; its weights follow the current stage's)
    int n_t = first, n_ks = 0;
    Geom nxt = cur;
    i32x4_t rxn = image_rsrc(cur.b);
    bool have_next = false;
    unsigned woff_nxt = 0u;
    auto advance = [&]() {                  // nxt := the stage after nxt
      if (++n_ks == nks) {
        n_ks = 0;
        ++n_t;
        const int b = nxt.b;
        tile_next(nxt);
        if (nxt.b != b) rxn = image_rsrc(nxt.b);
      }
      have_next = n_t < end;
      woff_nxt = stage_woff(nxt.nt, n_ks);
    };
    advance();
    // slab k (0 <= k < 2 NSLAB, counted from the current stage's slab 0) -> its byte offset in the weight image
    auto slab_woff = [&](int k) { return k < NSLAB ? woff_cur + (unsigned)(k * L::WSTAGE) : woff_nxt + (unsigned)((k - NSLAB) * L::WSTAGE); };
    static_assert(2 + 2 * DW <= NSLAB, "weight look-ahead stays within the next stage");
#pragma unroll
    for (int d = 0; d < DW; ++d)
#pragma unroll
      for (int i = 0; i < W_LOADS; ++i) ld16(rw4[(d + 2) % DW][i], rw, wv[i] + slab_woff(d + 2), 0u);
    __syncthreads();
    int wb2 = 2 * L::WSTAGE;               // LDS weight buffer of slab j+2
    int pboff = 0;                         // patch buffer of the current stage (byte offset)
    // loads of slab s, in issue order: W_LOADS weight loads, then nl(s) patch loads (granules s GPL .. of the next stage's patch)
    auto nl = [](int s) { s = ((s % NSLAB) + NSLAB) % NSLAB; const int r = P_LOADS - s * GPL; return r < 0 ? 0 : (r > GPL ? GPL : r); };
    const int nstages = (end - first) * nks;
    int q = 0;
    do {                                    // one K stage per iteration (at least one: first < end)
      const Stage z = stage_scalars(nxt);
      const i32x4_t rx = have_next ? rxn : rsrc_words(p.x, 0);       // (no next stage: a descriptor of size 0, every load out of range)
      const unsigned soff = z.tb + (unsigned)(n_ks * CS * 64);
      const unsigned ksoff = (unsigned)(n_ks * CS * 64);
#pragma unroll
      for (int s = 0; s < NSLAB; ++s) {
        // ---- weights: slab s+2 -> LDS, slab s+2+DW -> registers
        {
          int yw = nl(s - DW);
#pragma unroll
          for (int k = s - DW + 1; k < s; ++k) yw += W_LOADS + nl(k);
          f32x4 (&wset)[W_LOADS] = rw4[s % DW];
#pragma unroll
          for (int i = 0; i < W_LOADS; ++i) sp_wait_vm(wset[i], yw);
          const int a = wl0 + wb2;
#pragma unroll
          for (int i = 0; i < W_LOADS; ++i)
            if (!(HRSEG_WS_EXP & 8) && (i + 1 < W_LOADS || wlast)) *reinterpret_cast<f32x4*>(lds + a + i * 4096) = wset[i];
          const unsigned wo = slab_woff(s + 2 + DW);
#pragma unroll
          for (int i = 0; i < W_LOADS; ++i) ld16(wset[i], rw, (HRSEG_WS_EXP & 2) ? HRSEG_BUF_OOB : wv[i], wo);     // (slab offset: scalar)
        }
        // ---- patch of the next stage: the granules loaded at slab s-DP are stored, nl(s) granules are loaded
        if (s >= DP && nl(s - DP) > 0) {
          int yp = W_LOADS;
#pragma unroll
          for (int k = s - DP + 1; k < s; ++k) yp += W_LOADS + nl(k);
#pragma unroll
          for (int e = 0; e < GPL; ++e)
            if (e < nl(s - DP)) sp_wait_vm(pg[(s - DP) * GPL + e], yp + nl(s - DP) - 1 - e);
          if (have_next) {
#pragma unroll
            for (int e = 0; e < GPL; ++e)
              if (e < nl(s - DP)) patch_store1(pg[(s - DP) * GPL + e], (s - DP) * GPL + e, L::PATCH - pboff);
          }
        }
#pragma unroll
        for (int e = 0; e < GPL; ++e) {
          if (e < nl(s)) {
            const int i = s * GPL + e;
            if (HRSEG_WS_EXP & 4) ld16(pg[i], rx, HRSEG_BUF_OOB, 0u);
            else if (z.fast) ld16(pg[i], rx, rel[i], soff);
            else ld16(pg[i], rx, patch_voff(z, i), ksoff);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the LDS stores; the loads in flight stay in flight)
        stamped_barrier();
        wb2 = (wb2 == 2 * L::WSTAGE) ? 0 : wb2 + L::WSTAGE;
      }
      pboff = L::PATCH - pboff;
      woff_cur = woff_nxt;
      advance();
    } while (++q < nstages);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead loads of slabs past the block's last
  }
  if (stat) {
    // every consumer has added its last tile (LDS atomics complete before the barrier); the block's sums go out as row
    // `block_row` of the partial buffer [rows][2][N], which hrseg_bn_finalize adds up in row order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    double* row = p.stat_partial + (size_t)block_row * 2 * p.N;
    for (int i = tid; i < 2 * p.N; i += 512) row[i] = lstat[i];
  }
}

template <int NS, int TH, int WTN, int CS, int FLIP>
__global__ __launch_bounds__(512) void igemm_patch_ws_kernel(IgemmArgs p, int ntotal) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpPatchWsLds<NS, TH, WTN, CS>::BYTES];
  const int chunk = (ntotal + gridDim.x - 1) / gridDim.x;
  const int first = blockIdx.x * chunk;
  igemm_patch_ws_body<NS, TH, WTN, CS, FLIP>(p, lds, first, min(first + chunk, ntotal), blockIdx.x);
}

// grouped form: every problem runs the wave-specialised body on its own range of persistent blocks (grp.tiles[g]
// blocks for problem g, grp.ksplit[g] = its tile count), with its own channel tiling: kind 1 = 48 channels x 48-channel
// K stages, 2 = 96 x 48, 3 = 64 x 64, 4 = 48 x 48 on 16-row tiles
template <int NS, int FLIP>
__global__ __launch_bounds__(512) void igemm_patch_ws_group_kernel(IgemmGroup grp) {
  static_assert(SpPatchWsLds<NS, 16, 3, 3>::BYTES >= SpPatchWsLds<NS, 8, 4, 4>::BYTES &&
                SpPatchWsLds<NS, 16, 3, 3>::BYTES >= SpPatchWsLds<NS, 8, 6, 3>::BYTES, "LDS of the largest variant");
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpPatchWsLds<NS, 16, 3, 3>::BYTES];
  int gi = 0;
  while (gi + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[gi]) ++gi;
  const int local = blockIdx.x - (gi ? grp.blk_end[gi - 1] : 0);
  const int nblk = grp.tiles[gi], ntotal = grp.ksplit[gi];
  const int chunk = (ntotal + nblk - 1) / nblk;
  const int first = local * chunk, end = min(first + chunk, ntotal);
  const int kind = grp.kind[gi];
  if (kind == 1) igemm_patch_ws_body<NS, 8, 3, 3, FLIP>(grp.a[gi], lds, first, end, local);
  else if (kind == 2) igemm_patch_ws_body<NS, 8, 6, 3, FLIP>(grp.a[gi], lds, first, end, local);
  else if (kind == 3) igemm_patch_ws_body<NS, 8, 4, 4, FLIP>(grp.a[gi], lds, first, end, local);
  else igemm_patch_ws_body<NS, 16, 3, 3, FLIP>(grp.a[gi], lds, first, end, local);
}

// grouped launch whose problems run either body (the parallel HRNet branches: the wide high-resolution
// branches take the halo-patch body, the small low-resolution ones the im2col body with split-K)
template <int NS, int WTM, int WTN, int CS, int FLIP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sp_patch_min_waves(NS, WTN, CS), 2)))
void igemm_sp_pgroup_kernel(IgemmGroup grp) {
  constexpr int A = SpPatchLds<NS, 8, WTN, CS>::BYTES, B = SpLds<NS, WTN>::BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[A > B ? A : B];
  int gi = 0;
  while (gi + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[gi]) ++gi;
  const int local = blockIdx.x - (gi ? grp.blk_end[gi - 1] : 0);
  const int tiles = grp.tiles[gi];
  if (grp.kind[gi]) igemm_patch_sp_body<NS, 8, WTN, CS, FLIP>(grp.a[gi], lds, local, 1 << 29, tiles);   // one tile per block
  else igemm_sp_body<NS, WTM, WTN>(grp.a[gi], lds, local % tiles, tiles, local / tiles, grp.ksplit[gi]);
}

// --------------------------------------------------------------------------- weight gradient, 3x3 stride 1: all nine taps per block
// The tap-per-block weight gradient above pulls dy and x through L2 nine times and splits every element nine
// times.  Here a block owns a (16*TNK couts) x (16*TNK cins) tile of dW for ALL nine taps and walks a chunk of
// 4 x 16 pixel tiles: per tile it stages the dy tile (64 pixels) and the (4+2) x 18 x patch ONCE (split on the
// fly, bf16 pieces, [pixel][channel] images) and its three waves -- wave = kernel row kh -- read transposed
// fragments for their three taps kw from the same patch at shifted pixel positions.  The block keeps its
// 3 x TNK x TNK accumulator tiles per wave in registers over the whole chunk and writes them with PLAIN stores
// into its own slab of a workspace [chunk][Cout][9][Cin]; wgrad9_reduce_kernel then adds the chunks to dW in
// index order: no atomics, no cross-wave reduction, and the gradient is bit-reproducible run to run.
template <int NS, int TNK>
struct SpWgrad9Lds {
  static constexpr int S = sp_row_stride(16 * TNK);     // bytes per pixel row (both images)
  static constexpr int DYPIX = 64, XPIX = 6 * 18;
  static constexpr int PIECE = (DYPIX + XPIX) * S;
  static constexpr int BYTES = sp_np(NS) * PIECE;
};

struct Wgrad9Args {
  const float* x; const float* dy; float* ws;     // ws: [nchunks][Cout][9][Cin]
  const float* dymax;                              // device scalar |dy|_max (fp16x2 scaling) or null
  int ldx, lddy, B, H, W, Cin, Cout;
  int tiles_x, tiles_y, ntiles, nchunks, per;      // per = tiles per chunk
  int exp_nosplit;                                 // MEASUREMENT ONLY (hrseg_tune exp_nosplit_x): stage x as if it came pre-split
  int x_presplit;                                  // x is stored pre-split (hrseg_conv_shape_t.x_split)
};

template <int NS, int TNK>
__device__ __forceinline__ void wgrad9_sp_body(const Wgrad9Args& p, unsigned char* lds, const int pair, const int chunk) {
  using L = SpWgrad9Lds<NS, TNK>;
  constexpr int S = L::S, PIECE = L::PIECE, XBASE = L::DYPIX * S;
  constexpr int GPP = TNK * 4;                               // 16-byte granules per pixel
  // a round of the block's 192 threads stages PR whole pixels (thread -> pixel tid / GPP of the round, granule
  // tid % GPP of the pixel): the dy tile takes 64 / PR rounds -- one tile row each when PR = 16 -- and the x patch
  // the rest; no division by a runtime value and one small multiply per granule (a wave issues an instruction every
  // four cycles at best, and with 1.5 waves per SIMD the staging arithmetic is paid in MFMA time)
  constexpr int NT = 192, PR = NT / GPP;
  constexpr int DY_LOADS = (L::DYPIX + PR - 1) / PR, X_LOADS = (L::XPIX + PR - 1) / PR, LOADS = DY_LOADS + X_LOADS;
  const int tid = threadIdx.x, lane = tid & 63, kh = tid >> 6;     // wave = kernel row
  const int g = lane >> 4, li = lane & 15;
  const int nkt = p.Cin / (16 * TNK);
  const int ct = pair / nkt, kt = pair - ct * nkt;
  const int n0 = ct * 16 * TNK, k0 = kt * 16 * TNK;
  const int t_lo = chunk * p.per, t_hi = min(t_lo + p.per, p.ntiles);
  float dyscale, dyinv;                     // fp16x2: the gradient operand is scaled by 2^14 / 2^floor(log2 |max|)
  sp_pow2_scale(p.dymax, dyscale, dyinv);

  f32x4 rg[LOADS];
  const int pix0 = tid / GPP, gq = tid - pix0 * GPP;
  const bool swork = pix0 < PR;                                // (TNK = 3: all 192 threads; TNK = 4: 12 x 16)
  // What a thread's granules are does not depend on the tile: their byte offsets from the tile origin (dy) / the patch origin
  // (x: row y0-1, column x0-1) and their (row, column) there are computed once per block.  A tile that lies inside the image
  // with its halo loads them with these as the vector offset and the tile origin as the SCALAR offset -- no vector
  // instruction per granule; a border tile pays two range tests.  (Per tile this was three integer divisions and, per
  // granule, two quarter-rate 32-bit multiplies: with 1.5 waves per SIMD all of it is paid in MFMA time.)
  unsigned g_rel[LOADS];
  int g_yx[LOADS];                                             // (row << 8) | column; a row no image reaches where the granule does not exist
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    if (i < DY_LOADS) {
      const int pix = pix0 + PR * i;
      const bool ex = swork & (pix < L::DYPIX);
      g_rel[i] = ex ? ((unsigned)((pix >> 4) * p.W + (pix & 15)) * (unsigned)p.lddy + (unsigned)(n0 + 4 * gq)) * 4u : HRSEG_BUF_OOB;
      g_yx[i] = ex ? ((pix >> 4) << 8) | (pix & 15) : 0x400000;
    } else {
      const int pix = pix0 + PR * (i - DY_LOADS);
      const int py = (pix * 3641) >> 16, px = pix - py * 18;         // pix / 18
      const bool ex = swork & (pix < L::XPIX);
      g_rel[i] = ex ? ((unsigned)(py * p.W + px) * (unsigned)p.ldx + (unsigned)(k0 + 4 * gq)) * 4u : HRSEG_BUF_OOB;
      g_yx[i] = ex ? (py << 8) | px : 0x400000;
    }
  }
  // tile cursor of the NEXT load (tiles walk columns, rows, images): one division per block
  int c_tx = t_lo % p.tiles_x, c_ty = (t_lo / p.tiles_x) % p.tiles_y, c_b = (t_lo / p.tiles_x) / p.tiles_y;
  c_tx = __builtin_amdgcn_readfirstlane(c_tx); c_ty = __builtin_amdgcn_readfirstlane(c_ty); c_b = __builtin_amdgcn_readfirstlane(c_b);
  const unsigned lddy4 = (unsigned)p.lddy * 4u, ldx4 = (unsigned)p.ldx * 4u;
  auto tile_load = [&]() {
    const int y0 = c_ty * 4, x0 = c_tx * 16;
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy + (size_t)c_b * p.H * p.W * p.lddy, (size_t)p.H * p.W * p.lddy * 4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)c_b * p.H * p.W * p.ldx, (size_t)p.H * p.W * p.ldx * 4);
    const unsigned dyb = (unsigned)(y0 * p.W + x0) * lddy4;
    const unsigned xb = (unsigned)((y0 - 1) * p.W + x0 - 1) * ldx4;          // (mod 2^32 on a border tile: only added to offsets that exist)
    const bool dy_full = (y0 + 4 <= p.H) & (x0 + 16 <= p.W);
    const bool x_full = (y0 >= 1) & (x0 >= 1) & (y0 + 5 <= p.H) & (x0 + 17 <= p.W);
    if (dy_full) {
#pragma unroll
      for (int i = 0; i < DY_LOADS; ++i) rg[i] = buf_load4(rdy, g_rel[i], (int)dyb);
    } else {
#pragma unroll
      for (int i = 0; i < DY_LOADS; ++i) {
        const bool ok = (y0 + (g_yx[i] >> 8) < p.H) & (x0 + (g_yx[i] & 255) < p.W);
        rg[i] = buf_load4(rdy, ok ? g_rel[i] + dyb : HRSEG_BUF_OOB, 0);
      }
    }
    if (x_full) {
#pragma unroll
      for (int i = DY_LOADS; i < LOADS; ++i) rg[i] = buf_load4(rx, g_rel[i], (int)xb);
    } else {
#pragma unroll
      for (int i = DY_LOADS; i < LOADS; ++i) {
        const bool ok = ((unsigned)(y0 - 1 + (g_yx[i] >> 8)) < (unsigned)p.H) & ((unsigned)(x0 - 1 + (g_yx[i] & 255)) < (unsigned)p.W);
        rg[i] = buf_load4(rx, ok ? g_rel[i] + xb : HRSEG_BUF_OOB, 0);
      }
    }
    if (++c_tx == p.tiles_x) {
      c_tx = 0;
      if (++c_ty == p.tiles_y) { c_ty = 0; ++c_b; }
    }
  };
  auto tile_store = [&]() {
    // both images are [pixel][S bytes], the x patch behind the dy tile
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      u32x2 pc[sp_np(NS)];
      if (i >= DY_LOADS && (p.x_presplit | p.exp_nosplit)) {      // x stored pre-split (hrseg_conv_shape_t.x_split): the 16 bytes ARE {hi01, hi23, lo01, lo23}
        // (exp_nosplit: the same copy on fp32 data, wrong values on purpose: the ceiling measurement of tools/)
        const u32x4 raw = __builtin_bit_cast(u32x4, rg[i]);
#pragma unroll
        for (int s = 0; s < sp_np(NS); ++s) pc[s] = u32x2{raw[(2 * s) & 3], raw[(2 * s + 1) & 3]};
      } else {
        sp_split4<NS>(rg[i], pc, i < DY_LOADS ? dyscale : 1.f);
      }
      const int pix = (i < DY_LOADS) ? pix0 + PR * i : L::DYPIX + pix0 + PR * (i - DY_LOADS);
      const int o = pix * S + gq * 8;
      if (swork && (i < DY_LOADS ? pix < L::DYPIX : pix < L::DYPIX + L::XPIX)) {
#pragma unroll
        for (int s = 0; s < sp_np(NS); ++s) *reinterpret_cast<u32x2*>(lds + s * PIECE + o) = pc[s];
      }
    }
  };

  f32x4 acc[3][TNK][TNK];
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int n = 0; n < TNK; ++n)
#pragma unroll
      for (int k = 0; k < TNK; ++k) acc[w][n][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read lane offsets: lane 16g+i supplies row (i>>2) of a 4-pixel block, columns 4(i&3)..+3
  const int lrow = li >> 2, lcol = (li & 3) * 8;
  const int dy_lane = (4 * g + lrow) * S + lcol;                              // + (2ks+h)*16*S + n*32
  const int x_lane = XBASE + (kh * 18 + 4 * g + lrow) * S + lcol;             // + ((2ks+h)*18 + kw)*S + k*32

  if (t_lo < t_hi) tile_load();
  for (int t = t_lo; t < t_hi; ++t) {
    __syncthreads();                         // every wave is done with the previous tile's images
    tile_store();
    if (t + 1 < t_hi) tile_load();           // in flight behind this tile's MFMAs
    __syncthreads();
    // Six groups (pixel half ks, kernel column kw) of TNK x TNK tiles.  Inside a group the products go output-channel
    // block k outermost, so the x fragments of block k are dead after its TNK * products MFMAs and the reads of the
    // NEXT group's block k can be issued into the same registers right there, behind the MFMAs still to come (pinned
    // with sched_barrier: left alone the compiler bursts a group's reads in front of its MFMAs and every group starts
    // with an exposed LDS round trip).  Only the dy fragments of the second pixel half are read in the open.
    bf16x8 afr[TNK][sp_np(NS)], bfr[TNK][sp_np(NS)];
    auto read_a = [&](int ks) {
#pragma unroll
      for (int n = 0; n < TNK; ++n)
#pragma unroll
        for (int pc = 0; pc < sp_np(NS); ++pc) {
          const s16x4 v0 = sp_tr_read(lds + pc * PIECE + dy_lane + (2 * ks) * 16 * S + n * 32);
          const s16x4 v1 = sp_tr_read(lds + pc * PIECE + dy_lane + (2 * ks + 1) * 16 * S + n * 32);
          afr[n][pc] = __builtin_bit_cast(bf16x8, (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
        }
    };
    auto read_b = [&](int grp, int k) {
      const int ks = grp / 3, kw = grp % 3;
#pragma unroll
      for (int pc = 0; pc < sp_np(NS); ++pc) {
        const s16x4 v0 = sp_tr_read(lds + pc * PIECE + x_lane + ((2 * ks) * 18 + kw) * S + k * 32);
        const s16x4 v1 = sp_tr_read(lds + pc * PIECE + x_lane + ((2 * ks + 1) * 18 + kw) * S + k * 32);
        bfr[k][pc] = __builtin_bit_cast(bf16x8, (s16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
      }
    };
    read_a(0);
#pragma unroll
    for (int k = 0; k < TNK; ++k) read_b(0, k);
#pragma unroll
    for (int grp = 0; grp < 6; ++grp) {
      const int kw = grp % 3;
      if (grp == 3) read_a(1);
#pragma unroll
      for (int k = 0; k < TNK; ++k) {
#pragma unroll
        for (int pr = 0; pr < sp_nprod(NS); ++pr)
#pragma unroll
          for (int n = 0; n < TNK; ++n) acc[kw][n][k] = sp_mma_p<NS>(pr, afr[n], bfr[k], acc[kw][n][k]);
        __builtin_amdgcn_sched_barrier(0);
        if (grp + 1 < 6) read_b(grp + 1, k);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // this block's slab of the workspace: plain stores, every element written by exactly one lane
  float* out = p.ws + (size_t)chunk * p.Cout * 9 * p.Cin;
  const int row9 = 9 * p.Cin;
  const int obase = ((n0 + 4 * g) * 9 + kh * 3) * p.Cin + k0 + li;     // D row = 4*(lane>>4)+reg, D col = lane&15
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int n = 0; n < TNK; ++n)
#pragma unroll
      for (int k = 0; k < TNK; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          out[obase + (16 * n + e) * row9 + kw * p.Cin + 16 * k] = NS == 4 ? acc[kw][n][k][e] * dyinv : acc[kw][n][k][e];
}

#define WG9_MAXG 8
struct Wgrad9Group {
  int n;
  int blk_end[WG9_MAXG];
  Wgrad9Args a[WG9_MAXG];
};
template <int NS, int TNK>
__device__ __forceinline__ void wgrad9_sp_group_entry(const Wgrad9Group& grp, unsigned char* lds) {
  const int bid = (int)blockIdx.x;
  int gi = 0;
  while (gi + 1 < grp.n && bid >= grp.blk_end[gi]) ++gi;
  const int local = bid - (gi ? grp.blk_end[gi - 1] : 0);
  const Wgrad9Args& p = grp.a[gi];
  const int npairs = (p.Cout / (16 * TNK)) * (p.Cin / (16 * TNK));
  wgrad9_sp_body<NS, TNK>(p, lds, local % npairs, local / npairs);     // the tile pairs of one chunk are neighbours: same pixels
}
// 48-channel tiles: capped at 256 registers so that two blocks (six waves) share a CU
template <int NS>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad9_sp_group_kernel3(Wgrad9Group grp) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpWgrad9Lds<NS, 3>::BYTES];
  wgrad9_sp_group_entry<NS, 3>(grp, lds);
}
// 64-channel tiles: 3 x 16 accumulator tiles per wave, one block per CU
template <int NS>
__global__ __launch_bounds__(192) void wgrad9_sp_group_kernel4(Wgrad9Group grp) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[SpWgrad9Lds<NS, 4>::BYTES];
  wgrad9_sp_group_entry<NS, 4>(grp, lds);
}

// dW[i] += sum over chunks (in chunk order) of ws[chunk][i]; n4 = elements / 4 per problem
struct Wgrad9Reduce {
  int n;
  int blk_end[WG9_MAXG];
  const float* ws[WG9_MAXG];
  float* dw[WG9_MAXG];
  int nchunks[WG9_MAXG];
  long n4[WG9_MAXG];
};
#ifdef HRSEG_TU_WGRAD_SP
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(Wgrad9Reduce r) {
  // block = 32 consecutive float4 x 8 chunk groups (group j sums chunks j, j+8, ... in order); the eight partial
  // sums meet in LDS and are added in a fixed tree order: the same bits every run
  __shared__ f32x4 part[8][32];
  int gi = 0;
  while (gi + 1 < r.n && (int)blockIdx.x >= r.blk_end[gi]) ++gi;
  const int lo = gi ? r.blk_end[gi - 1] : 0;
  const long n4 = r.n4[gi];
  const f32x4* ws = reinterpret_cast<const f32x4*>(r.ws[gi]);
  f32x4* dw = reinterpret_cast<f32x4*>(r.dw[gi]);
  const int nch = r.nchunks[gi];
  const int e = threadIdx.x & 31, j = threadIdx.x >> 5;
  for (long i0 = (long)(blockIdx.x - lo) * 32; i0 < n4; i0 += (long)(r.blk_end[gi] - lo) * 32) {
    const long i = i0 + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
      for (int c = j; c < nch; c += 8) s += ws[(size_t)c * n4 + i];
    part[j][e] = s;
    __syncthreads();
    if (j == 0 && i < n4)
      dw[i] += ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) + ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e]));
    __syncthreads();
  }
}
#endif
