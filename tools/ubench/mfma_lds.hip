// MFMA (16x16x4 f32) fed by ds_read_b128 fragments: does the LDS-read + MFMA loop alone sustain the peak?
// Variants: reads per 24 MFMAs, waves per SIMD, conflict-free vs 80-byte-row layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NX, int NW, int STRIDE>   // NX pixel-tile frags, NW weight frags per tap; row stride in floats
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 1e-5f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fh = lane >> 4;
  f32x4 acc[2][NW][NX];
  for (int a = 0; a < 2; ++a) for (int n = 0; n < NW; ++n) for (int m = 0; m < NX; ++m) acc[a][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      f32x4 xf[NX], wf[NW];
#pragma unroll
      for (int m = 0; m < NX; ++m) xf[m] = *reinterpret_cast<const f32x4*>(lds + (((wave * NX + m + t / 3) * 18 + t % 3 + fi) * STRIDE + 4 * fh) % 8192);
#pragma unroll
      for (int n = 0; n < NW; ++n) wf[n] = *reinterpret_cast<const f32x4*>(lds + 8192 + ((t * 48 + 16 * n + fi) * 16 + 4 * (fh ^ ((-(fi >> 2)) & 3))) % 8192);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
          for (int m = 0; m < NX; ++m) acc[kk & 1][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][kk], xf[m][kk], acc[kk & 1][n][m], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int n = 0; n < NW; ++n) for (int m = 0; m < NX; ++m) s += acc[a][n][m][0] + acc[a][n][m][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f, int n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / n;
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 200;
  for (int bpc = 1; bpc <= 3; ++bpc) {
    int blocks = 256 * bpc;
    double fl = (double)blocks * 4 * iters * 9 * 4 * 2048.0;
    float ms = timeit([&] { hipLaunchKernelGGL((k<2, 3, 20>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("2x3 tiles, 80B patch rows   %d blocks/CU: %.1f TF\n", bpc, fl * 6 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k<2, 3, 16>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("2x3 tiles, 64B patch rows   %d blocks/CU: %.1f TF\n", bpc, fl * 6 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k<4, 3, 20>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("4x3 tiles, 80B patch rows   %d blocks/CU: %.1f TF\n", bpc, fl * 12 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k<1, 3, 20>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("1x3 tiles, 80B patch rows   %d blocks/CU: %.1f TF\n", bpc, fl * 3 / ms / 1e9);
  }
  return 0;
}
