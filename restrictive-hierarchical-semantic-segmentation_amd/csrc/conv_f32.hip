// Implicit GEMM on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32: forward / data gradient
// (see conv.hip for the family overview).
#include "conv_common.h"

// WTM x WTN 16x16 tiles per wave (4 waves split the pixel tile), KC 16-channel chunks per stage.
// LDS stage image: A [KC][BM][16 floats], B [KC][BN][16 floats]; two stages (double buffer).
// DB = LDS buffers: 2 = double buffer (one barrier per stage), 1 = single buffer (two barriers,
// half the LDS, so more blocks per CU).  gridDim.y > 1 = split-K over the stage list: partial
// sums are added with fp32 atomics.
template <int WTM, int WTN, int KC, int DB>
__device__ __forceinline__ void igemm_body(const IgemmArgs& p, float* lds, const int bid, const int nblk,
                                           const int ks_idx, const int ks_n) {
  constexpr int BM = 64 * WTM;  // pixels per block
  constexpr int BN = 16 * WTN;  // channels per block
  constexpr int A_ROWS = BM / 64;
  constexpr int B_F4 = BN * 4 * KC;
  constexpr int B_LOADS = (B_F4 + 255) / 256;
  constexpr int STAGE = (BM + BN) * 16 * KC;  // floats per LDS buffer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = p.N / BN;
  const int wg = xcd_remap(bid, nblk);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;

  // stage range of this block (split-K)
  const int kchunks = p.K / (16 * KC);
  const int nstages_all = p.ntaps * kchunks;
  const int per = (nstages_all + ks_n - 1) / ks_n;
  const int s_lo = ks_idx * per;
  const int s_hi = min(s_lo + per, nstages_all);
  const int nstages = s_hi - s_lo;

  // buffer resources: the input is addressed relative to the first image this tile touches, so the
  // 32-bit lane offsets only have to span the tile's own images (host-checked), not the tensor
  const int hw = p.Ho * p.Wo;
  const int b0 = m0 / hw;
  const __amdgpu_buffer_rsrc_t rx =
      make_rsrc(p.x + (size_t)b0 * p.Hi * p.Wi * p.ldx, (size_t)(p.B - b0) * p.Hi * p.Wi * p.ldx * 4);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (size_t)p.N * p.T * p.K * 4);

  // rows this thread stages: r = (tid>>2) + 64*i, 16-byte slot q = tid&3
  const int q = tid & 3;
  int rpix[A_ROWS], riy[A_ROWS], rix[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + (tid >> 2) + 64 * i;
    if (m < p.M) {
      const int b = fdiv(m, hw, p.rcp_hw);
      const int rem = m - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      rpix[i] = (b - b0) * p.Hi * p.Wi;
      riy[i] = oy * p.sy;
      rix[i] = ox * p.sx;
    } else {
      rpix[i] = 0;
      riy[i] = -(1 << 20);
      rix[i] = 0;
    }
  }
  // LDS store offsets (constant per thread)
  int a_st[A_ROWS], b_st[B_LOADS], b_row[B_LOADS], b_col[B_LOADS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int r = (tid >> 2) + 64 * i;
    a_st[i] = r * 16 + 4 * lds_slot(r, q);
  }
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int f = tid + 256 * i;
    const int j = f / (BN * 4), rem = f - j * (BN * 4);
    const int r = rem >> 2, qq = rem & 3;
    b_row[i] = r;
    b_col[i] = 16 * j + 4 * qq;
    b_st[i] = BM * 16 * KC + (j * BN + r) * 16 + 4 * lds_slot(r, qq);
  }

  // current tap / chunk
  int t = s_lo / kchunks, c = s_lo - t * kchunks;
  unsigned aoff[A_ROWS], boff[B_LOADS];   // byte offsets into rx / rw; HRSEG_BUF_OOB reads zeros
  auto set_tap = [&](int tap) {
    const int oy = (int)((p.offy_pk >> (4 * tap)) & 15) - 8;
    const int ox = (int)((p.offx_pk >> (4 * tap)) & 15) - 8;
    const int wt = (int)((p.wtap_pk >> (4 * tap)) & 15);
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int iy = riy[i] + oy, ix = rix[i] + ox;
      const bool ok = (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      aoff[i] = ok ? ((unsigned)(rpix[i] + iy * p.Wi + ix) * (unsigned)p.ldx + 4u * q) * 4u : HRSEG_BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      boff[i] = (tid + 256 * i < B_F4)
                    ? ((unsigned)((n0 + b_row[i]) * p.T + wt) * (unsigned)p.K + (unsigned)b_col[i]) * 4u
                    : HRSEG_BUF_OOB;
  };

  f32x4 ra[A_ROWS][KC], rb[B_LOADS];
  auto stage_load = [&]() {   // loads stage (t, c), then advances (t, c)
    const int c0 = c * 16 * KC;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i)
#pragma unroll
      for (int j = 0; j < KC; ++j) ra[i][j] = buf_load4(rx, aoff[i], (c0 + 16 * j) * 4);
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) rb[i] = buf_load4(rw, boff[i], c0 * 4);
    if (++c == kchunks) {
      c = 0;
      ++t;
      if (t < p.ntaps) set_tap(t);
    }
  };
  auto stage_store = [&](int buf) {
    float* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i)
#pragma unroll
      for (int j = 0; j < KC; ++j) *reinterpret_cast<f32x4*>(base + j * BM * 16 + a_st[i]) = ra[i][j];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      if (tid + 256 * i < B_F4) *reinterpret_cast<f32x4*>(base + b_st[i]) = rb[i];
  };

  // v_mfma_f32_16x16x4_f32 only reaches its issue rate with ~12 independent accumulators in
  // flight (measured: 4 chains 95 TF, 12 chains 150 TF, tools/ubench/mfma_peak.hip), so the k-steps
  // of a chunk go to KP separate partial accumulators per output tile, summed in the epilogue.
  constexpr int KP = (WTM * WTN <= 3) ? 4 : (WTM * WTN <= 6) ? 2 : 1;
  f32x4 acc[KP][WTN][WTM];
#pragma unroll
  for (int kp = 0; kp < KP; ++kp)
#pragma unroll
    for (int n = 0; n < WTN; ++n)
#pragma unroll
      for (int m = 0; m < WTM; ++m) acc[kp][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offset of this lane inside a 16-row tile
  const int frow = lane & 15;
  const int foff = frow * 16 + 4 * lds_slot(frow, lane >> 4);

  if (nstages > 0) {
    set_tap(t);
    stage_load();
    stage_store(0);
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const bool more = s + 1 < nstages;
    if (more) stage_load();
    const float* base = lds + ((DB == 2) ? (s & 1) : 0) * STAGE;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      f32x4 xf[WTM], wf[WTN];
#pragma unroll
      for (int m = 0; m < WTM; ++m)
        xf[m] = *reinterpret_cast<const f32x4*>(base + (j * BM + wave * 16 * WTM + 16 * m) * 16 + foff);
#pragma unroll
      for (int n = 0; n < WTN; ++n)
        wf[n] = *reinterpret_cast<const f32x4*>(base + BM * 16 * KC + (j * BN + 16 * n) * 16 + foff);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int n = 0; n < WTN; ++n)
#pragma unroll
          for (int m = 0; m < WTM; ++m)
            acc[k % KP][n][m] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[k % KP][n][m], 0, 0, 0);
    }
    if (DB == 1) __syncthreads();  // every wave is done reading before the buffer is rewritten
    if (more) stage_store((DB == 2) ? ((s + 1) & 1) : 0);
    __syncthreads();
  }

  // epilogue: lane holds channels n0+16n+4g..+3 of pixel row (lane&15)
  const int g = lane >> 4;
  const bool split = ks_n > 1;
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + (lane & 15);
    if (row >= p.M) continue;
    size_t pix = row;
    if (!p.direct_out) {
      const int b = fdiv(row, hw, p.rcp_hw);
      const int rem = row - b * hw;
      const int oy = fdiv(rem, p.Wo, p.rcp_w), ox = rem - oy * p.Wo;
      pix = (size_t)(b * p.Hy + oy * p.oys + p.oy0) * p.Wy + ox * p.oxs + p.ox0;
    }
    float* yrow = p.y + pix * p.ldy;
#pragma unroll
    for (int n = 0; n < WTN; ++n) {
      const int ch = n0 + 16 * n + 4 * g;
      f32x4 v = acc[0][n][m];
#pragma unroll
      for (int kp = 1; kp < KP; ++kp) v += acc[kp][n][m];
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

template <int WTM, int WTN, int KC, int DB>
__global__ __launch_bounds__(256) void igemm_conv_kernel(IgemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[DB * (64 * WTM + 16 * WTN) * 16 * KC];
  igemm_body<WTM, WTN, KC, DB>(p, lds, blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}

// FULL3X3 is a call-site tag only (same code): groups of full 3x3 stride-1 problems -- the parallel branch
// convs, forward and data-gradient, the dominant launches of a step -- get their own kernel symbol, so
// profiles list them apart from the small fuse-path / parity-class groups.
template <int WTM, int WTN, int KC, int DB, bool FULL3X3>
__global__ __launch_bounds__(256) void igemm_group_kernel(IgemmGroup grp) {
  __shared__ __attribute__((aligned(16))) float lds[DB * (64 * WTM + 16 * WTN) * 16 * KC];
  int g = 0;
  while (g + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[g]) ++g;
  const int local = blockIdx.x - (g ? grp.blk_end[g - 1] : 0);
  const int tiles = grp.tiles[g];
  igemm_body<WTM, WTN, KC, DB>(grp.a[g], lds, local % tiles, tiles, local / tiles, grp.ksplit[g]);
}

template <int WTM, int WTN, int KC, int DB>
static void launch_igemm(const IgemmArgs& a, int ksplit, hipStream_t st) {
  constexpr int BM = 64 * WTM, BN = 16 * WTN;
  const int grid = ceil_div(a.M, BM) * (a.N / BN);
  hipLaunchKernelGGL((igemm_conv_kernel<WTM, WTN, KC, DB>), dim3(grid, ksplit), dim3(256), 0, st, a);
}
template <int WTM, int WTN, int KC, int DB>
static void launch_igemm_group(const IgemmGroup& g, hipStream_t st) {
  bool full = true;
  for (int i = 0; i < g.n; ++i) full = full && g.a[i].ntaps == 9 && g.a[i].T == 9 && g.a[i].sy == 1 && g.a[i].oys == 1;
  if (full) hipLaunchKernelGGL((igemm_group_kernel<WTM, WTN, KC, DB, true>), dim3(g.blk_end[g.n - 1]), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((igemm_group_kernel<WTM, WTN, KC, DB, false>), dim3(g.blk_end[g.n - 1]), dim3(256), 0, st, g);
}

int launch_igemm_f32(const IgemmArgs& a, const IgemmPlan& pl, hipStream_t st) {
#define IG4(M_, N_, K_, D_) \
  if (pl.wtm == M_ && pl.wtn == N_ && pl.kc == K_ && pl.db == D_) { launch_igemm<M_, N_, K_, D_>(a, pl.ksplit, st); return 0; }
#define IG3(M_, N_, K_) IG4(M_, N_, K_, 1) IG4(M_, N_, K_, 2)
#define IG2(M_, N_) IG3(M_, N_, 1) IG3(M_, N_, 2) IG3(M_, N_, 3)
#define IG1(M_) IG2(M_, 1) IG2(M_, 2) IG2(M_, 3) IG2(M_, 4) IG2(M_, 6)
  IG1(1) IG1(2) IG1(4)
#undef IG1
#undef IG2
#undef IG3
#undef IG4
  return 1;
}

int launch_igemm_group_f32(const IgemmGroup& g, int wtm, int wtn, int kc, hipStream_t st) {
  if (wtm == 2 && wtn == 3 && kc == 3) launch_igemm_group<2, 3, 3, 1>(g, st);
  else if (wtm == 2 && wtn == 3) launch_igemm_group<2, 3, 1, 1>(g, st);
  else if (wtm == 2) return 1;
  else if (wtn == 3 && kc == 3) launch_igemm_group<1, 3, 3, 1>(g, st);
  else if (wtn == 3 && kc == 2) launch_igemm_group<1, 3, 2, 1>(g, st);
  else if (wtn == 3) launch_igemm_group<1, 3, 1, 1>(g, st);
  else if (kc == 3) launch_igemm_group<1, 4, 3, 1>(g, st);
  else if (kc == 2) launch_igemm_group<1, 4, 2, 1>(g, st);
  else launch_igemm_group<1, 4, 1, 1>(g, st);
  return 0;
}
