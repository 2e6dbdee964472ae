// Weight gradient on the exact-fp32 matrix instruction (see conv.hip for the family overview).
#include "conv_common.h"

// --------------------------------------------------------------------------- wgrad on MFMA

// Block = (tap, 16*TN couts, 16*TK cins, pixel range).  PIX pixels per LDS stage; the 4 waves split
// the stage's pixels (the MFMA k dimension), so each wave accumulates the full TN x TK tile set and
// the block reduces across waves through LDS before the atomic add.
// Staging: thread (r = tid>>2, q = tid&3) owns pixel rows r, r+64 and the 16-byte slot q of every
// 16-channel chunk, so the pixel -> (b,oy,ox) decode is done once per row per stage.
template <int TN, int TK, int PIX, int DB>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, float* lds, const int bx, int id) {
  constexpr int SA = 16 * TN + ((TN % 2) ? 0 : 16);  // row strides with (stride % 32) == 16
  constexpr int SB = 16 * TK + ((TK % 2) ? 0 : 16);
  constexpr int ROWS = PIX / 64;                     // rows per thread
  constexpr int STAGE = PIX * (SA + SB);
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

  // buffer resources relative to this block's pixel range (dy) / its first image (x): rows past the
  // range and padding pixels read zeros through the descriptor's range check
  const int hw = p.Ho * p.Wo;
  const int b_lo = lo / hw;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy + (size_t)lo * p.lddy, (size_t)max(hi - lo, 0) * p.lddy * 4);
  const __amdgpu_buffer_rsrc_t rx =
      make_rsrc(p.x + (size_t)b_lo * p.Hi * p.Wi * p.ldx, (size_t)(p.B - b_lo) * p.Hi * p.Wi * p.ldx * 4);

  f32x4 ra[ROWS][TN], rb[ROWS][TK];
  auto stage_load = [&](int s) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int ml = s * PIX + r0 + 64 * i;          // row inside the block's range
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
  auto stage_store = [&](int buf) {
    float* a = lds + buf * STAGE;
    float* b = a + PIX * SA;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int r = r0 + 64 * i;
#pragma unroll
      for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(a + r * SA + 16 * j + 4 * q) = ra[i][j];
#pragma unroll
      for (int j = 0; j < TK; ++j) *reinterpret_cast<f32x4*>(b + r * SB + 16 * j + 4 * q) = rb[i][j];
    }
  };

  constexpr int KP = (TN * TK <= 9) ? 2 : 1;     // independent accumulation chains, see igemm_body
  f32x4 acc2[KP][TN][TK];
#pragma unroll
  for (int kp = 0; kp < KP; ++kp)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int k = 0; k < TK; ++k) acc2[kp][n][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nstages > 0) {
    stage_load(0);
    stage_store(0);
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const bool more = s + 1 < nstages;
    if (more) stage_load(s + 1);
    const float* a = lds + ((DB == 2) ? (s & 1) : 0) * STAGE;
    const float* b = a + PIX * SA;
#pragma unroll
    for (int ks4 = 0; ks4 < PIX / 16; ++ks4) {
      const int row = wave * (PIX / 4) + ks4 * 4 + (lane >> 4);
      float af[TN], bf[TK];
#pragma unroll
      for (int n = 0; n < TN; ++n) af[n] = a[row * SA + 16 * n + (lane & 15)];
#pragma unroll
      for (int k = 0; k < TK; ++k) bf[k] = b[row * SB + 16 * k + (lane & 15)];
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int k = 0; k < TK; ++k)
          acc2[ks4 % KP][n][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[n], bf[k], acc2[ks4 % KP][n][k], 0, 0, 0);
    }
    if (DB == 1) __syncthreads();
    if (more) stage_store((DB == 2) ? ((s + 1) & 1) : 0);
    __syncthreads();
  }

  // cross-wave reduction: red[wave][tile][r*64 + lane]
#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      f32x4 v = acc2[0][n][k];
#pragma unroll
      for (int kp = 1; kp < KP; ++kp) v += acc2[kp][n][k];
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[((wave * TN + n) * TK + k) * 256 + r * 64 + lane] = v[r];
    }
  __syncthreads();
  const int r = tid >> 6, l = tid & 63;
#pragma unroll
  for (int n = 0; n < TN; ++n)
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) v += lds[((wv * TN + n) * TK + k) * 256 + tid];
      const int co = n0 + 16 * n + 4 * (l >> 4) + r;  // D row = 4*(lane>>4)+reg
      const int ci = k0 + 16 * k + (l & 15);          // D col = lane&15
      atomicAdd(p.dw + ((size_t)co * p.T + tap) * p.Cin + ci, v);
    }
}


template <int TN, int TK, int PIX, int DB>
struct WgradLds {
  static constexpr int SA = 16 * TN + ((TN % 2) ? 0 : 16), SB = 16 * TK + ((TK % 2) ? 0 : 16);
  static constexpr int STAGE = PIX * (SA + SB), RED = 4 * TN * TK * 256;
  static constexpr int FLOATS = (DB * STAGE > RED) ? DB * STAGE : RED;
};
// Block order: the (tap, tile) blocks of ONE pixel range are consecutive in the XCD-remapped id, so
// the taps that re-read the same dy / x rows run together on one XCD and hit its L2 (the PMC
// counters showed 3.2x the algorithmic bytes fetched with the pixel range as the fast index).
template <int TN, int TK, int PIX, int DB>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[WgradLds<TN, TK, PIX, DB>::FLOATS];
  const int tiles = gridDim.y, nblk = gridDim.x * gridDim.y;
  const int r = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, nblk);
  wgrad_body<TN, TK, PIX, DB>(p, lds, r / tiles, r % tiles);
}
template <int TN, int TK, int PIX, int DB>
__global__ __launch_bounds__(256) void wgrad_group_kernel(WgradGroup grp) {
  __shared__ __attribute__((aligned(16))) float lds[WgradLds<TN, TK, PIX, DB>::FLOATS];
  int g = 0;
  while (g + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[g]) ++g;
  const int lo = g ? grp.blk_end[g - 1] : 0;
  const int nblk = grp.blk_end[g] - lo;
  const int tiles = nblk / grp.gx[g];
  const int r = xcd_remap(blockIdx.x - lo, nblk);
  wgrad_body<TN, TK, PIX, DB>(grp.a[g], lds, r / tiles, r % tiles);
}

template <int TN, int TK, int PIX, int DB>
static int launch_wgrad_cfg(WgradArgs a, int target_blocks, hipStream_t st) {
  const int tiles = (a.Cout / (16 * TN)) * (a.Cin / (16 * TK)) * a.T;
  int ksplit = target_blocks / tiles;
  if (ksplit < 1 || hrseg_g_deterministic) ksplit = 1;       // deterministic: one pixel range, one adder per element
  int ppb = ceil_div(ceil_div(a.M, ksplit), PIX) * PIX;
  if (ppb < 4 * PIX) ppb = 4 * PIX;
  a.pix_per_block = ppb;
  if (int e = check_wgrad_span(a)) return e;
  const int gx = ceil_div(a.M, ppb);
  hipLaunchKernelGGL((wgrad_kernel<TN, TK, PIX, DB>), dim3(gx, tiles), dim3(256), 0, st, a);
  return 0;
}

template <int TN, int TK>
static int launch_wgrad(const WgradArgs& a, int pix, int db, int target, hipStream_t st) {
  if (pix == 64 && db == 1) return launch_wgrad_cfg<TN, TK, 64, 1>(a, target, st);
  if (pix == 64) return launch_wgrad_cfg<TN, TK, 64, 2>(a, target, st);
  if (db == 1) return launch_wgrad_cfg<TN, TK, 128, 1>(a, target, st);
  return launch_wgrad_cfg<TN, TK, 128, 2>(a, target, st);
}

int launch_wgrad_f32(const WgradArgs& a, int tn, int tk, int pix, int db, int target, hipStream_t st) {
#define WG(TN_, TK_) if (tn == TN_ && tk == TK_) return launch_wgrad<TN_, TK_>(a, pix, db, target, st);
  WG(1, 1) WG(1, 2) WG(1, 3) WG(1, 4) WG(2, 1) WG(2, 2) WG(2, 3) WG(2, 4)
  WG(3, 1) WG(3, 2) WG(3, 3) WG(3, 4) WG(4, 1) WG(4, 2) WG(4, 3) WG(4, 4)
#undef WG
  return HRSEG_ERR_UNSUPPORTED;
}

int launch_wgrad_group_f32(const WgradGroup& g, int tn, int tk, int nblocks, hipStream_t st) {
#define WGG(TN_, TK_) if (tn == TN_ && tk == TK_) { hipLaunchKernelGGL((wgrad_group_kernel<TN_, TK_, 64, 1>), dim3(nblocks), dim3(256), 0, st, g); return 0; }
  WGG(3, 3) WGG(3, 4) WGG(4, 3) WGG(4, 4)
#undef WGG
  return 1;
}
