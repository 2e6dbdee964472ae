// Split-precision weight gradients (tap-per-block and nine-tap): kernel instances + launchers.
#define HRSEG_TU_WGRAD_SP
#include "conv_common.h"
#include "conv_sp.h"


template <int NS>
static int launch_wgrad_sp(const WgradArgs& a, int tn, int tk, int gx, int tiles, hipStream_t st) {
#define WS(TN_, TK_) if (tn == TN_ && tk == TK_) { hipLaunchKernelGGL((wgrad_sp_kernel<NS, TN_, TK_>), dim3(gx, tiles), dim3(256), 0, st, a); return 0; }
  WS(1, 1) WS(1, 2) WS(1, 3) WS(1, 4) WS(2, 1) WS(2, 2) WS(2, 3) WS(2, 4)
  WS(3, 1) WS(3, 2) WS(3, 3) WS(3, 4) WS(4, 1) WS(4, 2) WS(4, 3) WS(4, 4)
  if constexpr (NS == 4) { WS(5, 5) }       // 80 x 80 tile (80 KB of LDS, one block per CU): the 720-channel layer
#undef WS
  return HRSEG_ERR_UNSUPPORTED;
}
int launch_wgrad_sp_kernel(int ns, const WgradArgs& a, int tn, int tk, int gx, int tiles, hipStream_t st) {
  return ns == 4 ? launch_wgrad_sp<4>(a, tn, tk, gx, tiles, st) : ns == 3 ? launch_wgrad_sp<3>(a, tn, tk, gx, tiles, st)
       : ns == 2 ? launch_wgrad_sp<2>(a, tn, tk, gx, tiles, st) : launch_wgrad_sp<1>(a, tn, tk, gx, tiles, st);
}
int launch_wgrad_spw_kernel(const WgradArgs& a, int gx, int tiles, hipStream_t st) {       // fp16x2, 240 x 144 block tiles
  hipLaunchKernelGGL((wgrad_spw_kernel<4, 5, 3, 3, 3>), dim3(gx, tiles), dim3(576), 0, st, a);
  return 0;
}
int launch_wgrad_group_sp(const WgradGroup& g, int tn, int tk, int nblocks, hipStream_t st) {      // fp16x2 only
#define WGS(TN_, TK_) if (tn == TN_ && tk == TK_) { hipLaunchKernelGGL((wgrad_sp_group_kernel<4, TN_, TK_>), dim3(nblocks), dim3(256), 0, st, g); return 0; }
  WGS(3, 3) WGS(3, 4) WGS(4, 3) WGS(4, 4)
#undef WGS
  return 1;
}
int launch_wgrad9_kernels(int ns, int tnk, const Wgrad9Group& g, int nblocks, const Wgrad9Reduce& r, int rblocks, hipStream_t st) {
#define W9(NS_) if (ns == NS_) { \
    if (tnk == 3) hipLaunchKernelGGL((wgrad9_sp_group_kernel3<NS_>), dim3(nblocks), dim3(192), 0, st, g); \
    else hipLaunchKernelGGL((wgrad9_sp_group_kernel4<NS_>), dim3(nblocks), dim3(192), 0, st, g); }
  W9(1) W9(2) W9(3) W9(4)
#undef W9
  HRSEG_LAUNCH_CHECK("wgrad9");
  hipLaunchKernelGGL(wgrad9_reduce_kernel, dim3(rblocks), dim3(256), 0, st, r);
  HRSEG_LAUNCH_CHECK("wgrad9_reduce");
  return 0;
}
