// Split-precision grouped launch whose problems run either the halo-patch or the im2col body: kernel instances + launcher.
#include "conv_common.h"
#include "conv_sp.h"


template <int NS>
static int launch_pgroup(const IgemmGroup& g, int wtm, int wtn, int cs, int flip, hipStream_t st) {
  const dim3 grid(g.blk_end[g.n - 1]);
#define SPP(M_, N_, C_) \
  if (wtm == M_ && wtn == N_ && cs == C_) { \
    if (flip) hipLaunchKernelGGL((igemm_sp_pgroup_kernel<NS, M_, N_, C_, 1>), grid, dim3(256), 0, st, g); \
    else hipLaunchKernelGGL((igemm_sp_pgroup_kernel<NS, M_, N_, C_, 0>), grid, dim3(256), 0, st, g); \
    return 0; }
  SPP(1, 3, 3) SPP(2, 3, 3) SPP(1, 4, 4) SPP(2, 4, 4)
#undef SPP
  return 1;
}
int launch_sp_pgroup_kernel(int ns, const IgemmGroup& g, int wtm, int wtn, int cs, int flip, hipStream_t st) {
  return ns == 4 ? launch_pgroup<4>(g, wtm, wtn, cs, flip, st) : ns == 3 ? launch_pgroup<3>(g, wtm, wtn, cs, flip, st)
       : ns == 2 ? launch_pgroup<2>(g, wtm, wtn, cs, flip, st) : launch_pgroup<1>(g, wtm, wtn, cs, flip, st);
}
