// Where does an igemm stage spend its cycles?  Diagnostic build: the production kernel body with
// s_memtime stamps per phase (results of this build are timing shares, never product output).
// hipcc --offload-arch=gfx950 -O3 -I../../restrictive-hierarchical-semantic-segmentation_amd/csrc igemm_lab.hip -o igemm_lab
#include "../../restrictive-hierarchical-semantic-segmentation_amd/csrc/error.hip"
#include "../../restrictive-hierarchical-semantic-segmentation_amd/csrc/conv.hip"
#include <vector>
#include <stdlib.h>

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// copy of igemm_body<2,3,1,1> structure with stamps: phases = [load-issue, frag-read+wait, mfma, sync1, vm-wait+store, sync2]
template <int WTM, int WTN>
__global__ __launch_bounds__(256) void igemm_stamped(IgemmArgs p, unsigned long long* prof) {
  constexpr int KC = 1, DB = 1;
  constexpr int BM = 64 * WTM, BN = 16 * WTN, A_ROWS = BM / 64, B_F4 = BN * 4 * KC, B_LOADS = (B_F4 + 255) / 256;
  constexpr int STAGE = (BM + BN) * 16 * KC;
  __shared__ __attribute__((aligned(16))) float lds[DB * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = p.N / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
  const int kchunks = p.K / 16, nstages = p.ntaps * kchunks;
  const int q = tid & 3;
  int rpix[A_ROWS], riy[A_ROWS], rix[A_ROWS];
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + (tid >> 2) + 64 * i;
    if (m < p.M) { const int b = m / (p.Ho * p.Wo); const int rem = m - b * (p.Ho * p.Wo); const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      rpix[i] = b * p.Hi * p.Wi; riy[i] = oy * p.sy; rix[i] = ox * p.sx; } else { rpix[i] = 0; riy[i] = -(1 << 20); rix[i] = 0; }
  }
  int a_st[A_ROWS], b_st[B_LOADS], b_row[B_LOADS], b_col[B_LOADS];
  for (int i = 0; i < A_ROWS; ++i) { const int r = (tid >> 2) + 64 * i; a_st[i] = r * 16 + 4 * lds_slot(r, q); }
  for (int i = 0; i < B_LOADS; ++i) { const int f = tid + 256 * i; const int r = f >> 2, qq = f & 3; b_row[i] = r; b_col[i] = 4 * qq; b_st[i] = BM * 16 + r * 16 + 4 * lds_slot(r, qq); }
  int t = 0, c = 0;
  const float* aptr[A_ROWS]; bool aok[A_ROWS]; const float* bptr[B_LOADS];
  auto set_tap = [&](int tap) {
    const int oy = (int)((p.offy_pk >> (4 * tap)) & 15) - 8, ox = (int)((p.offx_pk >> (4 * tap)) & 15) - 8, wt = (int)((p.wtap_pk >> (4 * tap)) & 15);
    for (int i = 0; i < A_ROWS; ++i) { const int iy = riy[i] + oy, ix = rix[i] + ox; aok[i] = (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      aptr[i] = p.x + (size_t)(aok[i] ? (rpix[i] + iy * p.Wi + ix) : 0) * p.ldx + 4 * q; }
    for (int i = 0; i < B_LOADS; ++i) bptr[i] = p.w + ((size_t)(n0 + b_row[i]) * p.T + wt) * p.K + b_col[i];
  };
  f32x4 ra[A_ROWS], rb[B_LOADS];
  auto stage_load = [&]() {
    const int c0 = c * 16;
    for (int i = 0; i < A_ROWS; ++i) { f32x4 v = {0.f, 0.f, 0.f, 0.f}; if (aok[i]) v = *reinterpret_cast<const f32x4*>(aptr[i] + c0); ra[i] = v; }
    for (int i = 0; i < B_LOADS; ++i) { f32x4 v = {0.f, 0.f, 0.f, 0.f}; if (tid + 256 * i < B_F4) v = *reinterpret_cast<const f32x4*>(bptr[i] + c0); rb[i] = v; }
    if (++c == kchunks) { c = 0; ++t; if (t < p.ntaps) set_tap(t); }
  };
  auto stage_store = [&]() {
    for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<f32x4*>(lds + a_st[i]) = ra[i];
    for (int i = 0; i < B_LOADS; ++i) if (tid + 256 * i < B_F4) *reinterpret_cast<f32x4*>(lds + b_st[i]) = rb[i];
  };
  f32x4 acc[WTN][WTM];
  for (int n = 0; n < WTN; ++n) for (int m = 0; m < WTM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, foff = frow * 16 + 4 * lds_slot(frow, lane >> 4);
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  set_tap(0); stage_load(); stage_store(); __syncthreads();
  const unsigned long long tstart = stamp();
  for (int s = 0; s < nstages; ++s) {
    const bool more = s + 1 < nstages;
    unsigned long long t0 = stamp();
    if (more) stage_load();
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = stamp();
    f32x4 xf[WTM], wf[WTN];
    for (int m = 0; m < WTM; ++m) xf[m] = *reinterpret_cast<const f32x4*>(lds + (wave * 16 * WTM + 16 * m) * 16 + foff);
    for (int n = 0; n < WTN; ++n) wf[n] = *reinterpret_cast<const f32x4*>(lds + BM * 16 + (16 * n) * 16 + foff);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t2 = stamp();
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int n = 0; n < WTN; ++n)
#pragma unroll
        for (int m = 0; m < WTM; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[n][m], 0, 0, 0);
    // make the stamp wait for the MFMAs: read one accumulator element
    asm volatile("" ::"v"(acc[WTN - 1][WTM - 1][0]));
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t3 = stamp();
    __syncthreads();
    unsigned long long t4 = stamp();
    if (more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stage_store(); }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t5 = stamp();
    __syncthreads();
    unsigned long long t6 = stamp();
    ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5;
  }
  const unsigned long long tend = stamp();
  const int g = lane >> 4;
  for (int m = 0; m < WTM; ++m) {
    const int row = m0 + wave * 16 * WTM + 16 * m + (lane & 15);
    if (row >= p.M) continue;
    float* yrow = p.y + (size_t)row * p.ldy;
    for (int n = 0; n < WTN; ++n) *reinterpret_cast<f32x4*>(yrow + n0 + 16 * n + 4 * g) = acc[n][m];
  }
  if (lane == 0) {
    unsigned long long* o = prof + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 6; ++i) o[i] = ph[i];
    o[6] = tend - tstart;
    o[7] = tstart;
  }
}

int main() {
  const int B = 4, H = 155, C = 48;
  hrseg_conv_shape_t s{B, H, H, C, C, H, H, C, C, 3, 1};
  const size_t nx = (size_t)B * H * H * C, nw = (size_t)C * 9 * C;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 5e-5f;
  float *x, *w, *y, *y2; unsigned long long* prof;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, nx * 4); hipMalloc(&y2, nx * 4);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  IgemmArgs a; fill_fwd_args(a, x, w, nullptr, y2, &s);
  const int nblk = ceil_div(a.M, 128);
  hipMalloc(&prof, (size_t)nblk * 4 * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) hrseg_conv_fwd(x, w, nullptr, y, &s, nullptr);
  hipEventRecord(e0); for (int rep = 0; rep < 10; ++rep) hrseg_conv_fwd(x, w, nullptr, y, &s, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("production kernel: %.1f us\n", ms * 100);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((igemm_stamped<2, 3>), dim3(nblk), dim3(256), 0, 0, a, prof);
  hipEventRecord(e0); for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((igemm_stamped<2, 3>), dim3(nblk), dim3(256), 0, 0, a, prof); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); printf("stamped kernel:    %.1f us (stamps cost time; read the SHARES)\n", ms * 100);
  std::vector<unsigned long long> hp((size_t)nblk * 4 * 8);
  hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost);
  double tot[7] = {0}; unsigned long long tmin = ~0ull, tmax = 0;
  for (size_t i = 0; i < (size_t)nblk * 4; ++i) { for (int k = 0; k < 7; ++k) tot[k] += hp[i * 8 + k]; tmin = std::min(tmin, hp[i * 8 + 7]); tmax = std::max(tmax, hp[i * 8 + 7] + hp[i * 8 + 6]); }
  const char* names[6] = {"issue global loads", "LDS fragment reads + wait", "24 MFMAs (until result)", "barrier 1 (reads done)", "vmcnt wait + LDS store", "barrier 2 (stores visible)"};
  const double n = (double)nblk * 4 * 27;
  for (int k = 0; k < 6; ++k) printf("  %-28s %7.0f cycles/stage/wave  %5.1f%%\n", names[k], tot[k] / n, 100 * tot[k] / tot[6]);
  printf("  main loop per wave %.0f cycles; kernel span (first start -> last end) %.0f cycles (s_memtime ticks, 100 MHz => x24 for 2.4 GHz?)\n", tot[6] / (nblk * 4.0), (double)(tmax - tmin));
  std::vector<float> hy(nx), hy2(nx);
  hipMemcpy(hy.data(), y, nx * 4, hipMemcpyDeviceToHost); hipMemcpy(hy2.data(), y2, nx * 4, hipMemcpyDeviceToHost);
  double md = 0; for (size_t i = 0; i < nx; ++i) md = std::max(md, (double)fabsf(hy[i] - hy2[i]));
  printf("max |production - stamped| = %g\n", md);
  return 0;
}
