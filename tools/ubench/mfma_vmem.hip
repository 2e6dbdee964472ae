// How much MFMA rate does concurrent vector-memory traffic cost?  24 MFMAs (12 chains) + 5 ds_read_b128 per stage
// as in the conv kernel, plus NL global loads per stage (all L1/L2 hits), optionally direct-to-LDS.
// hipcc --offload-arch=gfx950 -O3 -w mfma_vmem.hip -o mfma_vmem     (diagnostic only)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NL, int MODE>   // MODE 0: dwordx4 to VGPR, 1: dword to VGPR, 2: dwordx4 direct to LDS, 3: dwordx4 to VGPR + ds_write
__device__ __forceinline__ void body(const float* __restrict__ x, float* out, int iters, float* lds) {
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-5f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fh = lane >> 4;
  const int foff = fi * 16 + 4 * (fh ^ ((-(fi >> 2)) & 3));
  f32x4 acc[2][3][2];
  for (int a = 0; a < 2; ++a) for (int n = 0; n < 3; ++n) for (int m = 0; m < 2; ++m) acc[a][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 ld[NL > 0 ? NL : 1];
  for (int i = 0; i < (NL > 0 ? NL : 1); ++i) ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  const float* g = x + ((blockIdx.x & 63) * 256 + threadIdx.x) * 4;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < NL; ++i) { sink += ld[i]; ld[i] = *reinterpret_cast<const f32x4*>(g + ((it + i) & 7) * 65536); }
      if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4*>(lds + 8192 + ((i * 256 + threadIdx.x) * 4 & 4095)) = sink;
      }
    } else if (MODE == 4) {     // SGPR base + 32-bit per-lane offset (global_load ... saddr)
      const unsigned off = ((blockIdx.x & 63) * 256 + threadIdx.x) * 16;
#pragma unroll
      for (int i = 0; i < NL; ++i) { sink += ld[i]; ld[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(x) + (size_t)(off + (unsigned)(((it + i) & 7) * 262144))); }
    } else if (MODE == 5) {     // buffer_load_dwordx4 with a resource descriptor, 32-bit voffset
      const unsigned off = ((blockIdx.x & 63) * 256 + threadIdx.x) * 16;
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        sink += ld[i];
        i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, ((it + i) & 7) * 262144, 0);
        ld[i] = __builtin_bit_cast(f32x4, v);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NL; ++i) { sink[0] += ld[i][0]; ld[i][0] = g[((it + i) & 7) * 65536]; }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) __builtin_amdgcn_global_load_lds(g + ((it + i) & 7) * 65536, lds + 8192 + (wave * NL + i) * 256 % 4096, 16, 0, 0);
    }
    f32x4 xf[2], wf[3];
#pragma unroll
    for (int m = 0; m < 2; ++m) xf[m] = *reinterpret_cast<const f32x4*>(lds + ((wave * 2 + m) * 256 + (it & 3) * 2048 + foff) % 8192);
#pragma unroll
    for (int n = 0; n < 3; ++n) wf[n] = *reinterpret_cast<const f32x4*>(lds + (n * 256 + ((it + 1) & 3) * 2048 + foff) % 8192);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[kk & 1][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][kk], xf[m][kk], acc[kk & 1][n][m], 0, 0, 0);
  }
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int i = 0; i < (NL > 0 ? NL : 1); ++i) s += ld[i][0] + ld[i][3];
  for (int a = 0; a < 2; ++a) for (int n = 0; n < 3; ++n) for (int m = 0; m < 2; ++m) s += acc[a][n][m][0] + acc[a][n][m][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + lds[8192 + threadIdx.x];
}
template <int NL, int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) float lds[8192 + 4096];
  body<NL, MODE>(x, out, iters, lds);
}
template <typename F> float timeit(F f, int n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / n;
}
template <int NL, int MODE> void go(const char* name, const float* x, float* out) {
  const int iters = 2000;
  printf("  %-30s loads/stage %d :", name, NL);
  for (int bpc = 1; bpc <= 4; ++bpc) {
    const int blocks = 256 * bpc;
    const double fl = (double)blocks * 4 * iters * 24 * 2048.0;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) { float ms = timeit([&] { hipLaunchKernelGGL((k<NL, MODE>), dim3(blocks), dim3(256), 0, 0, x, out, iters); }, 3); if (ms < best) best = ms; }
    printf("  %d/CU %6.1f TF", bpc, fl / best / 1e9);
  }
  printf("\n");
}
int main() {
  float *x, *out; (void)hipMalloc(&x, 8 * 65536 * 4 + 65536 * 4); (void)hipMalloc(&out, 4096 * 256 * 4);
  (void)hipMemset(x, 0, 8 * 65536 * 4 + 65536 * 4);
  go<0, 0>("no loads", x, out);
  go<1, 0>("dwordx4 -> VGPR", x, out);
  go<3, 0>("dwordx4 -> VGPR", x, out);
  go<6, 0>("dwordx4 -> VGPR", x, out);
  go<3, 1>("dword -> VGPR", x, out);
  go<3, 2>("dwordx4 -> LDS direct", x, out);
  go<6, 2>("dwordx4 -> LDS direct", x, out);
  go<3, 3>("dwordx4 -> VGPR -> ds_write", x, out);
  go<3, 4>("dwordx4 saddr+voffset32", x, out);
  go<6, 4>("dwordx4 saddr+voffset32", x, out);
  go<3, 5>("buffer_load_dwordx4", x, out);
  go<6, 5>("buffer_load_dwordx4", x, out);
  go<0, 0>("no loads (again)", x, out);
  return 0;
}
