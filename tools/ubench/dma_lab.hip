// Implicit-GEMM stage loop fed by direct-to-LDS loads (global_load_lds_dwordx4, gfx950): no VGPR
// round trip, no ds_write, NBUF-deep LDS ring, one barrier per stage.  Compared against the
// production register-staged kernel.  Diagnostic build.
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-result dma_lab.hip -o dma_lab
#include "../../restrictive-hierarchical-semantic-segmentation_amd/csrc/error.hip"
#include "../../restrictive-hierarchical-semantic-segmentation_amd/csrc/conv.hip"
#include <vector>
#include <stdlib.h>

__device__ __attribute__((aligned(64))) float g_zero_row[16];

#define VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0f70)

template <int WTM, int WTN, int KC, int NBUF>
__device__ __forceinline__ void igemm_dma_body(const IgemmArgs& p, float* lds) {
  constexpr int BM = 64 * WTM, BN = 16 * WTN;
  constexpr int STAGE = (BM + BN) * 16 * KC;          // floats
  constexpr int NB = KC * WTN;                        // weight chunks (16 rows x 64 B) per stage
  constexpr int NBW = (NB + 3) / 4;                   // per wave (waves beyond the count re-load the last chunk)
  constexpr int LPS = KC * WTM + NBW;                 // loads per wave per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = p.N / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
  const int kchunks = p.K / (16 * KC), nstages = p.ntaps * kchunks;
  const int r = lane >> 2;                                    // row inside a 16-row chunk
  const int q = (lane & 3) ^ ((-(r >> 2)) & 3);               // logical 16-byte slot that lands in physical slot lane&3
  int rpix[WTM], riy[WTM], rix[WTM];
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    const int m = m0 + wave * 16 * WTM + 16 * i + r;
    if (m < p.M) {
      const int b = m / (p.Ho * p.Wo); const int rem = m - b * (p.Ho * p.Wo); const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      rpix[i] = b * p.Hi * p.Wi; riy[i] = oy * p.sy; rix[i] = ox * p.sx;
    } else { rpix[i] = 0; riy[i] = -(1 << 20); rix[i] = 0; }
  }
  int bj[NBW], bn[NBW];                                       // this wave's weight chunks: k-sub-chunk j, 16-row tile n
#pragma unroll
  for (int i = 0; i < NBW; ++i) { int cidx = min(wave + 4 * i, NB - 1); bj[i] = cidx / WTN; bn[i] = cidx - bj[i] * WTN; }
  int t = 0, c = 0;
  const float* aptr[WTM]; bool aok[WTM]; const float* bptr[NBW];
  auto set_tap = [&](int tap) {
    const int oy = (int)((p.offy_pk >> (4 * tap)) & 15) - 8, ox = (int)((p.offx_pk >> (4 * tap)) & 15) - 8, wt = (int)((p.wtap_pk >> (4 * tap)) & 15);
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      const int iy = riy[i] + oy, ix = rix[i] + ox; aok[i] = (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      aptr[i] = p.x + (size_t)(rpix[i] + iy * p.Wi + ix) * p.ldx + 4 * q;
    }
#pragma unroll
    for (int i = 0; i < NBW; ++i) bptr[i] = p.w + ((size_t)(n0 + 16 * bn[i] + r) * p.T + wt) * p.K + 16 * bj[i] + 4 * q;
  };
  auto stage_issue = [&](int buf) {   // LPS direct-to-LDS loads of stage (t,c) into ring slot buf; stays on the last stage at the end
    float* base = lds + buf * STAGE;
    const int c0 = c * 16 * KC;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const float* src = aok[i] ? (aptr[i] + c0 + 16 * j) : (g_zero_row + 4 * (lane & 3));
        __builtin_amdgcn_global_load_lds(src, base + (j * BM + wave * 16 * WTM + 16 * i) * 16, 16, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < NBW; ++i)
      __builtin_amdgcn_global_load_lds(bptr[i] + c0, base + BM * 16 * KC + (bj[i] * BN + 16 * bn[i]) * 16, 16, 0, 0);
    if (t * kchunks + c + 1 < nstages) { if (++c == kchunks) { c = 0; ++t; set_tap(t); } }
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
  set_tap(0);
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b) stage_issue(b);          // stages 0 .. NBUF-2 in flight
  int buf = 0;
  for (int s = 0; s < nstages; ++s) {
    // stage s has landed when at most (NBUF-2) later stages of this wave are still in flight
    VMCNT((NBUF - 2) * LPS);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();       // bare barrier: __syncthreads() would add vmcnt(0) and drain the ring
    asm volatile("" ::: "memory");      // everyone's part of stage s is in LDS; everyone is done reading stage s-1
    int nb = buf + NBUF - 1; if (nb >= NBUF) nb -= NBUF;
    stage_issue(nb);                    // stage s+NBUF-1 into the slot stage s-1 occupied
    const float* base = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      f32x4 xf[WTM], wf[WTN];
#pragma unroll
      for (int m = 0; m < WTM; ++m) xf[m] = *reinterpret_cast<const f32x4*>(base + (j * BM + wave * 16 * WTM + 16 * m) * 16 + foff);
#pragma unroll
      for (int n = 0; n < WTN; ++n) wf[n] = *reinterpret_cast<const f32x4*>(base + BM * 16 * KC + (j * BN + 16 * n) * 16 + foff);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int n = 0; n < WTN; ++n)
#pragma unroll
          for (int m = 0; m < WTM; ++m)
            acc[k % KP][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[k % KP][n][m], 0, 0, 0);
    }
    if (++buf == NBUF) buf = 0;
  }
  VMCNT(0);
  const int g = lane >> 4;
#pragma unroll
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + (lane & 15);
    if (row >= p.M) continue;
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

template <int WTM, int WTN, int KC, int NBUF>
__global__ __launch_bounds__(256) void igemm_dma(IgemmArgs p) {
  __shared__ __attribute__((aligned(1024))) float lds[NBUF * (64 * WTM + 16 * WTN) * 16 * KC];
  igemm_dma_body<WTM, WTN, KC, NBUF>(p, lds);
}

static hipEvent_t e0, e1;
template <int WTM, int WTN, int KC, int NBUF>
static void run(const char* name, const IgemmArgs& a, const std::vector<float>& ref, float* y2, double flop) {
  const int nblk = ceil_div(a.M, 64 * WTM) * (a.N / (16 * WTN));
  hipMemset(y2, 0, ref.size() * 4);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((igemm_dma<WTM, WTN, KC, NBUF>), dim3(nblk), dim3(256), 0, 0, a);
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((igemm_dma<WTM, WTN, KC, NBUF>), dim3(nblk), dim3(256), 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h(ref.size());
  hipMemcpy(h.data(), y2, h.size() * 4, hipMemcpyDeviceToHost);
  double md = 0; for (size_t i = 0; i < h.size(); ++i) md = std::max(md, (double)fabsf(h[i] - ref[i]));
  printf("  %-26s %5d blocks  %7.1f us  %6.1f TF   maxdiff %g\n", name, nblk, ms * 50, flop / (ms * 50e-6) * 1e-12, md);
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
  run<2, 3, 1, 2>("128x48 kc1 ring2", a, ref, y2, flop);
  run<2, 3, 1, 3>("128x48 kc1 ring3", a, ref, y2, flop);
  run<2, 3, 1, 4>("128x48 kc1 ring4", a, ref, y2, flop);
  run<2, 3, 3, 2>("128x48 kc3 ring2", a, ref, y2, flop);
  run<2, 3, 3, 3>("128x48 kc3 ring3", a, ref, y2, flop);
  run<1, 3, 3, 2>("64x48 kc3 ring2", a, ref, y2, flop);
  run<1, 3, 3, 3>("64x48 kc3 ring3", a, ref, y2, flop);
  run<1, 3, 1, 3>("64x48 kc1 ring3", a, ref, y2, flop);
  run<4, 3, 1, 2>("256x48 kc1 ring2", a, ref, y2, flop);
  run<4, 3, 1, 3>("256x48 kc1 ring3", a, ref, y2, flop);
  hipFree(x); hipFree(w); hipFree(y); hipFree(y2);
}

int main() {
  hipEventCreate(&e0); hipEventCreate(&e1);
  shape(4, 155, 48);
  shape(4, 78, 96);
  shape(4, 310, 48);
  return 0;
}
