// Shared device/host helpers for the hrseg HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hrseg.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
