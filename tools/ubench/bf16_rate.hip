// bf16 matrix-pipe questions behind the split-precision conv kernels (csrc/conv_bf16.hip):
//  (1) v_mfma_f32_16x16x32_bf16 vs the K=16 form v_mfma_f32_16x16x16_bf16: same FLOP rate or half?
//  (2) an fp32 operand split on the fly into NS bf16 pieces (v_and / v_sub / v_perm, 5.5 VALU ops per
//      element for NS=3) next to the P = 1/3/6 products per tile: how much of the MFMA rate survives?
// hipcc --offload-arch=gfx950 -O3 bf16_rate.hip -o bf16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 au = {0x3f803f80u + threadIdx.x, 0x3f803f81u, 0x3f823f80u, 0x3f803f83u};
  u32x4 bu = {0x3f813f80u, 0x3f803f85u + threadIdx.x, 0x3f803f80u, 0x3f873f80u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  s16x4 a4 = {(short)au[0], (short)au[1], (short)au[2], (short)au[3]};
  s16x4 b4 = {(short)bu[0], (short)bu[1], (short)bu[2], (short)bu[3]};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// hi halves of two floats -> one dword of two bf16 (truncation)
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

// one wave = WTM pixel tiles x WTN channel tiles; per step: raw fp32 pixel fragments come from LDS (stand-in
// for the global loads), are split into NS pieces and multiplied with constant weight fragments.
template <int NS, int WTM, int WTN>
__global__ __launch_bounds__(256) void k_split(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float raw[256 * 8 + 64];
  for (int i = threadIdx.x; i < 256 * 8 + 64; i += 256) raw[i] = 1.0f + 1e-3f * i;
  __syncthreads();
  constexpr int P = NS == 1 ? 1 : NS == 2 ? 3 : 6;
  f32x4 acc[WTM][WTN];
  for (int m = 0; m < WTM; ++m) for (int n = 0; n < WTN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 wf[WTN][NS];
  for (int n = 0; n < WTN; ++n)
    for (int s = 0; s < NS; ++s) {
      u32x4 u = {0x3f803f80u + threadIdx.x + n, 0x3f803f81u + s, 0x3f823f80u, 0x3f803f83u};
      wf[n][s] = __builtin_bit_cast(bf16x8, u);
    }
  const volatile f32x4* src = reinterpret_cast<const volatile f32x4*>(raw + threadIdx.x * 8);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < WTM; ++m) {
      f32x4 r0 = const_cast<const f32x4*>(reinterpret_cast<const volatile f32x4*>(src))[(it + m) & 1];
      f32x4 r1 = const_cast<const f32x4*>(reinterpret_cast<const volatile f32x4*>(src))[((it + m) & 1) + 2];
      asm volatile("" : "+v"(r0), "+v"(r1));
      float x[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
      bf16x8 xf[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        u32x4 u;
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = pack_hi(x[2 * j], x[2 * j + 1]);
        xf[s] = __builtin_bit_cast(bf16x8, u);
        if (s + 1 < NS) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = x[j] - trunc_bf16(x[j]);
        }
      }
#pragma unroll
      for (int n = 0; n < WTN; ++n) {
        // products in order of decreasing weight: (0,0) (0,1) (1,0) (0,2) (2,0) (1,1)
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][0], xf[0], acc[m][n], 0, 0, 0);
        if (P >= 3) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][1], xf[0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][0], xf[1], acc[m][n], 0, 0, 0);
        }
        if (P >= 6) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][2], xf[0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][0], xf[2], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][1], xf[1], acc[m][n], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
  for (int m = 0; m < WTM; ++m) for (int n = 0; n < WTN; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float timeit(F f, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / n;
}

template <int NS, int WTM, int WTN> void run_split(float* out, int bpc) {
  const int iters = 2000, blocks = 256 * bpc;
  constexpr int P = NS == 1 ? 1 : NS == 2 ? 3 : 6;
  float ms = timeit([&] { hipLaunchKernelGGL((k_split<NS, WTM, WTN>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
  const double mf = (double)blocks * 4 * iters * WTM * WTN * P;           // MFMAs issued
  const double alg = (double)blocks * 4 * iters * WTM * WTN * 16384.0;    // fp32-equivalent FLOP
  printf("split NS=%d WTM=%d WTN=%d  %d blocks/CU: %.3f ms  matrix pipe %.0f TF (bf16)  fp32-equivalent %.1f TF  %.1f cyc/MFMA/SIMD@2.4GHz\n",
         NS, WTM, WTN, bpc, ms, mf * 16384 / ms / 1e9, alg / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / (256.0 * 4)) );
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000;
  for (int bpc = 1; bpc <= 2; bpc *= 2) {
    int blocks = 256 * bpc;
    float ms = timeit([&] { hipLaunchKernelGGL((k_mfma<0, 4>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("16x16x32 bf16 4acc %d blocks/CU: %.3f ms %.0f TF\n", bpc, ms, (double)blocks * 4 * iters * 4 * 16384 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k_mfma<0, 1>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("16x16x32 bf16 1acc %d blocks/CU: %.3f ms %.0f TF\n", bpc, ms, (double)blocks * 4 * iters * 1 * 16384 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k_mfma<1, 4>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("16x16x16 bf16 4acc %d blocks/CU: %.3f ms %.0f TF\n", bpc, ms, (double)blocks * 4 * iters * 4 * 8192 / ms / 1e9);
    run_split<1, 2, 3>(out, bpc);
    run_split<2, 2, 3>(out, bpc);
    run_split<3, 2, 3>(out, bpc);
    run_split<3, 4, 3>(out, bpc);
    run_split<2, 2, 6>(out, bpc);
    run_split<3, 2, 6>(out, bpc);
  }
  return 0;
}
