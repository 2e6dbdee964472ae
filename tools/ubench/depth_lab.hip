// Is the im2col conv kernel latency-bound on its global loads?  Variants of the production stage loop:
//   DEPTH = stages of global loads in flight (1 = production, 2 = two register sets)
//   HOT   = every block gathers the SAME pixels (all loads hit L1/L2): the memory system taken out
// Loads are unconditional (clamped address + select) so the wait counts stay static.
// hipcc --offload-arch=gfx950 -O3 depth_lab.hip -o depth_lab     (diagnostic only, never product output)
#include "../../restrictive-hierarchical-semantic-segmentation_amd/csrc/error.hip"
#include "../../restrictive-hierarchical-semantic-segmentation_amd/csrc/conv.hip"
#include <vector>
#include <stdlib.h>

template <int WTM, int WTN, int KC, int DEPTH, bool HOT, int FLAGS = 0>
__global__ __launch_bounds__(256) void igemm_var(IgemmArgs p) {
  constexpr int BM = 64 * WTM, BN = 16 * WTN, A_ROWS = BM / 64, B_F4 = BN * 4 * KC, B_LOADS = (B_F4 + 255) / 256;
  constexpr int STAGE = (BM + BN) * 16 * KC;
  __shared__ __attribute__((aligned(16))) float lds[STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = p.N / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
  const int kchunks = p.K / (16 * KC), nstages_real = p.ntaps * kchunks, nstages = nstages_real;
  const int nloop = (FLAGS & 16) ? 8 * nstages : nstages;
  const int q = tid & 3;
  int rpix[A_ROWS], riy[A_ROWS], rix[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = (HOT ? 0 : m0) + (tid >> 2) + 64 * i;
    if (m < p.M) {
      const int b = m / (p.Ho * p.Wo); const int rem = m - b * (p.Ho * p.Wo); const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      rpix[i] = b * p.Hi * p.Wi; riy[i] = oy * p.sy; rix[i] = ox * p.sx;
    } else { rpix[i] = 0; riy[i] = -(1 << 20); rix[i] = 0; }
  }
  int a_st[A_ROWS], b_st[B_LOADS], b_row[B_LOADS], b_col[B_LOADS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) { const int r = (tid >> 2) + 64 * i; a_st[i] = r * 16 + 4 * lds_slot(r, q); }
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int f = min(tid + 256 * i, B_F4 - 1);
    const int j = f / (BN * 4), rem = f - j * (BN * 4); const int r = rem >> 2, qq = rem & 3;
    b_row[i] = r; b_col[i] = 16 * j + 4 * qq; b_st[i] = BM * 16 * KC + (j * BN + r) * 16 + 4 * lds_slot(r, qq);
  }
  int t = 0, c = 0;
  const float* aptr[A_ROWS]; bool aok[A_ROWS]; const float* bptr[B_LOADS];
  unsigned aoff[A_ROWS], boff[B_LOADS];
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((size_t)p.B * p.Hi * p.Wi * p.ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((size_t)p.N * p.T * p.K * 4), 0x00020000);
  auto set_tap = [&](int tap) {
    const int oy = (int)((p.offy_pk >> (4 * tap)) & 15) - 8, ox = (int)((p.offx_pk >> (4 * tap)) & 15) - 8, wt = (int)((p.wtap_pk >> (4 * tap)) & 15);
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int iy = riy[i] + oy, ix = rix[i] + ox; aok[i] = (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      aptr[i] = p.x + (size_t)(aok[i] ? (rpix[i] + iy * p.Wi + ix) : 0) * p.ldx + 4 * q;
      aoff[i] = aok[i] ? (unsigned)(((rpix[i] + iy * p.Wi + ix) * p.ldx + 4 * q) * 4) : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      bptr[i] = p.w + ((size_t)(n0 + b_row[i]) * p.T + wt) * p.K + b_col[i];
      boff[i] = (unsigned)((((n0 + b_row[i]) * p.T + wt) * p.K + b_col[i]) * 4);
    }
  };
  struct Regs { f32x4 a[A_ROWS][KC]; f32x4 b[B_LOADS]; bool ok[A_ROWS]; };
  auto stage_load = [&](Regs& r) {   // unconditional loads of stage (t,c); stays on the last stage at the end
    const int c0 = c * 16 * KC;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      r.ok[i] = (FLAGS & 32) ? true : aok[i];
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        if (FLAGS & 32) r.a[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, aoff[i], (c0 + 16 * j) * 4, 0));
        else r.a[i][j] = *reinterpret_cast<const f32x4*>(aptr[i] + c0 + 16 * j);
      }
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      if (FLAGS & 32) r.b[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, boff[i], c0 * 4, 0));
      else r.b[i] = *reinterpret_cast<const f32x4*>(bptr[i] + c0);
    }
    if (t * kchunks + c + 1 < nstages) { if (++c == kchunks) { c = 0; ++t; set_tap(t); } }
  };
  auto stage_store = [&](const Regs& r) {
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i)
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        f32x4 v = r.a[i][j]; if (!r.ok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + j * BM * 16 + a_st[i]) = v;
      }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) if (tid + 256 * i < B_F4) *reinterpret_cast<f32x4*>(lds + b_st[i]) = r.b[i];
  };
  constexpr int KP = (WTM * WTN <= 3) ? 4 : (WTM * WTN <= 6) ? 2 : 1;
  f32x4 acc[KP][WTN][WTM];
#pragma unroll
  for (int kp = 0; kp < KP; ++kp)
#pragma unroll
    for (int n = 0; n < WTN; ++n)
#pragma unroll
      for (int m = 0; m < WTM; ++m) acc[kp][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, foff = frow * 16 + 4 * lds_slot(frow, lane >> 4);
  auto compute = [&]() {
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      f32x4 xf[WTM], wf[WTN];
#pragma unroll
      for (int m = 0; m < WTM; ++m) xf[m] = *reinterpret_cast<const f32x4*>(lds + (j * BM + wave * 16 * WTM + 16 * m) * 16 + foff);
#pragma unroll
      for (int n = 0; n < WTN; ++n) wf[n] = *reinterpret_cast<const f32x4*>(lds + BM * 16 * KC + (j * BN + 16 * n) * 16 + foff);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int n = 0; n < WTN; ++n)
#pragma unroll
          for (int m = 0; m < WTM; ++m)
            acc[k % KP][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[k % KP][n][m], 0, 0, 0);
    }
  };
  Regs R0, R1;
  set_tap(0);
  stage_load(R0); stage_store(R0); __syncthreads();
  stage_load(R0);                       // stage 1
  if (DEPTH == 2) stage_load(R1);       // stage 2
  // iteration s: LDS holds stage s, Ra holds stage s+1 (and Rb stage s+2)
  auto iter = [&](int s, Regs& Ra) {
    compute();
    if (!(FLAGS & 4)) __syncthreads();
    if (!(FLAGS & 2)) { if (s + 1 < nloop) stage_store(Ra); }
    if (!(FLAGS & 1)) stage_load(Ra);   // stage s+1+DEPTH (clamped to the last one)
    if (!(FLAGS & 4)) __syncthreads();
  };
  if (DEPTH == 1) {
    for (int s = 0; s < nloop; ++s) iter(s, R0);
  } else {
    for (int s = 0; s < nloop; s += 2) {
      iter(s, R0);
      if (s + 1 < nloop) iter(s + 1, R1);
    }
  }
  const int g = lane >> 4;
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + (lane & 15);
    if (row >= p.M) continue;
    if ((FLAGS & 8) && acc[0][0][m][0] != 12345.f) continue;
    float* yrow = p.y + (size_t)row * p.ldy;
#pragma unroll
    for (int n = 0; n < WTN; ++n) {
      f32x4 v = acc[0][n][m];
#pragma unroll
      for (int kp = 1; kp < KP; ++kp) v += acc[kp][n][m];
      *reinterpret_cast<f32x4*>(yrow + n0 + 16 * n + 4 * g) = v;
    }
  }
}

static hipEvent_t e0, e1;
template <int WTM, int WTN, int KC, int DEPTH, bool HOT, int FLAGS = 0>
static void run(const char* name, const IgemmArgs& a, const std::vector<float>& ref, float* y2, double flop) {
  const int nblk = ceil_div(a.M, 64 * WTM) * (a.N / (16 * WTN)) / ((FLAGS & 16) ? 8 : 1);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((igemm_var<WTM, WTN, KC, DEPTH, HOT, FLAGS>), dim3(nblk), dim3(256), 0, 0, a);
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((igemm_var<WTM, WTN, KC, DEPTH, HOT, FLAGS>), dim3(nblk), dim3(256), 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(ref.size());
  hipMemcpy(h.data(), y2, h.size() * 4, hipMemcpyDeviceToHost);
  double md = 0; for (size_t i = 0; i < h.size(); ++i) md = std::max(md, (double)fabsf(h[i] - ref[i]));
  printf("  %-34s %4d blocks  %7.1f us  %6.1f TF   maxdiff %s%g\n", name, nblk, ms * 50, flop / (ms * 50e-6) * 1e-12, HOT ? "(hot: n/a) " : "", md);
}

static void shape(int B, int H, int C) {
  hrseg_conv_shape_t s{B, H, H, C, C, H, H, C, C, 3, 1};
  const size_t nx = (size_t)B * H * H * C, nw = (size_t)C * 9 * C;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 5e-5f;
  float *x, *w, *y, *y2;
  hipMalloc(&x, nx * 4 + 4096); hipMalloc(&w, nw * 4 + 4096); hipMalloc(&y, nx * 4); hipMalloc(&y2, nx * 4);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  const double flop = 2.0 * B * H * H * C * 9.0 * C;
  for (int rep = 0; rep < 3; ++rep) hrseg_conv_fwd(x, w, nullptr, y, &s, nullptr);
  hipEventRecord(e0); for (int rep = 0; rep < 20; ++rep) hrseg_conv_fwd(x, w, nullptr, y, &s, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("shape B=%d %dx%d C=%d: production %.1f us %.1f TF\n", B, H, H, C, ms * 50, flop / (ms * 50e-6) * 1e-12);
  std::vector<float> ref(nx); hipMemcpy(ref.data(), y, nx * 4, hipMemcpyDeviceToHost);
  IgemmArgs a; fill_fwd_args(a, x, w, nullptr, y2, &s);
  run<2, 3, 1, 1, false>("128x48 kc1", a, ref, y2, flop);
  run<2, 3, 1, 1, false, 32>("128x48 kc1 buffer_load", a, ref, y2, flop);
  run<2, 3, 3, 1, false>("128x48 kc3", a, ref, y2, flop);
  run<2, 3, 3, 1, false, 32>("128x48 kc3 buffer_load", a, ref, y2, flop);
  run<1, 3, 3, 1, false>("64x48 kc3", a, ref, y2, flop);
  run<1, 3, 3, 1, false, 32>("64x48 kc3 buffer_load", a, ref, y2, flop);
  run<2, 6, 1, 1, false>("128x96 kc1", a, ref, y2, flop);
  run<2, 6, 1, 1, false, 32>("128x96 kc1 buffer_load", a, ref, y2, flop);
  hipFree(x); hipFree(w); hipFree(y); hipFree(y2);
}

int main() {
  hipEventCreate(&e0); hipEventCreate(&e1);
  shape(4, 155, 48);
  shape(4, 78, 96);
  shape(4, 310, 48);
  shape(4, 155, 96);
  return 0;
}
