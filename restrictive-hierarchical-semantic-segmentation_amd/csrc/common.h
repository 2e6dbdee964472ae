// Shared device/host helpers for the hrseg HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hrseg.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hrseg_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hrseg_f16x2 __attribute__((ext_vector_type(2)));
#ifdef __HIPCC__
// One fp32 granule (4 channels of a pixel) in PRE-SPLIT fp16x2 form: dwords {hi01, hi23, lo01, lo23} with hi = round-toward-zero
// fp16 of x and lo = round-to-nearest fp16 of (x - hi) -- exactly what conv_sp.h sp_split4<4> produces on the fly (scale 1), so a
// convolution that reads a tensor stored this way multiplies the same bits as one that splits the fp32 tensor itself.
__device__ __forceinline__ hrseg_u32x4 hrseg_split_f16x2(const f32x4& x) {
  const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
  const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[2], x[3]));
  const hrseg_f16x2 a = __builtin_bit_cast(hrseg_f16x2, h0), b = __builtin_bit_cast(hrseg_f16x2, h1);
  const hrseg_f16x2 l0 = {(_Float16)(x[0] - (float)a[0]), (_Float16)(x[1] - (float)a[1])};
  const hrseg_f16x2 l1 = {(_Float16)(x[2] - (float)b[0]), (_Float16)(x[3] - (float)b[1])};
  return hrseg_u32x4{h0, h1, __builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}
// ... and back: hi + lo (exact in fp32: the value the convolutions multiply, the fp32 original rounded to 22 bits)
__device__ __forceinline__ float hrseg_h2f(unsigned bits16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); }
__device__ __forceinline__ f32x4 hrseg_join_f16x2(const hrseg_u32x4& g) {
  const unsigned h01 = g[0], h23 = g[1], l01 = g[2], l23 = g[3];
  f32x4 r;
  r[0] = hrseg_h2f(h01 & 0xffffu) + hrseg_h2f(l01 & 0xffffu);
  r[1] = hrseg_h2f(h01 >> 16) + hrseg_h2f(l01 >> 16);
  r[2] = hrseg_h2f(h23 & 0xffffu) + hrseg_h2f(l23 & 0xffffu);
  r[3] = hrseg_h2f(h23 >> 16) + hrseg_h2f(l23 >> 16);
  return r;
}
#endif

#define HRSEG_WAVE 64

void hrseg_set_error(const char* fmt, ...);
// hrseg_tune("deterministic", 1): every reduction that would add floats with atomics in a run-dependent order
// takes its single-adder form instead (no split-K, one pixel range per weight-gradient tile, one block per image
// in the head / loss reductions): bit-reproducible gradients at some cost in speed
extern int hrseg_g_deterministic;

#define HRSEG_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      hrseg_set_error(__VA_ARGS__);           \
      return HRSEG_ERR_INVALID_ARG;           \
    }                                         \
  } while (0)

#define HRSEG_LAUNCH_CHECK(name)                                                  \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      hrseg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return HRSEG_ERR_LAUNCH;                                                    \
    }                                                                             \
  } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one).  Remap a
// linear block id so that each XCD owns a contiguous chunk of the work list
// (bijective for any grid size).
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
