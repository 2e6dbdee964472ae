// Split-precision implicit GEMM, block-synchronous halo-patch body (3x3 stride 1): kernel instances + launcher.
#include "conv_common.h"
#include "conv_sp.h"


template <int NS>
static int launch_patch_sp(const IgemmArgs& a, int wtn, int cs, int flip, int blocks, int ntotal, hipStream_t st) {
  const dim3 grid((unsigned)blocks);
#define PS(N_, C_) if (wtn == N_ && cs == C_) { \
    if (flip) hipLaunchKernelGGL((igemm_patch_sp_kernel<NS, 8, N_, C_, 1>), grid, dim3(256), 0, st, a, ntotal); \
    else hipLaunchKernelGGL((igemm_patch_sp_kernel<NS, 8, N_, C_, 0>), grid, dim3(256), 0, st, a, ntotal); \
    return 0; }
  PS(3, 3) PS(3, 4) PS(4, 3) PS(4, 4) PS(6, 3) PS(6, 4)
#undef PS
  return 1;
}
int launch_patch_sp_kernel(int ns, const IgemmArgs& a, int wtn, int cs, int flip, int blocks, int ntotal, hipStream_t st) {
  return ns == 4 ? launch_patch_sp<4>(a, wtn, cs, flip, blocks, ntotal, st) : ns == 3 ? launch_patch_sp<3>(a, wtn, cs, flip, blocks, ntotal, st)
       : ns == 2 ? launch_patch_sp<2>(a, wtn, cs, flip, blocks, ntotal, st) : launch_patch_sp<1>(a, wtn, cs, flip, blocks, ntotal, st);
}
