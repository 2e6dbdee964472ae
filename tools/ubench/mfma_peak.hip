// What does back-to-back fp32 MFMA sustain on this device, and at what clock?
// hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, unsigned long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / n;
}
int main() {
  float* out; unsigned long long* clk; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
  const int iters = 4000;
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    int blocks = 256 * bpc;
    float ms = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); }, 5);
    double fl = (double)blocks * 4 * iters * 4 * 2048;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("16x16x4 4acc  %d blocks/CU: %.3f ms %.1f TF  clock %.2f GHz\n", bpc, ms, fl / ms / 1e9, (double)h[0] / h[1] * 0.1);
    ms = timeit([&] { hipLaunchKernelGGL(k16<12>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); }, 5);
    fl = (double)blocks * 4 * iters * 12 * 2048;
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("16x16x4 12acc %d blocks/CU: %.3f ms %.1f TF  clock %.2f GHz\n", bpc, ms, fl / ms / 1e9, (double)h[0] / h[1] * 0.1);
    ms = timeit([&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    fl = (double)blocks * 4 * iters * 4 * 4096;
    printf("32x32x2 4acc  %d blocks/CU: %.3f ms %.1f TF\n", bpc, ms, fl / ms / 1e9);
  }
  return 0;
}
