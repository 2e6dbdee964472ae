// Wave-specialised fp16x2 / bf16 halo-patch kernels (3x3 stride 1) and the pre-split weight image kernel: instances + launchers.
#define HRSEG_TU_WS
#include "conv_common.h"
#include "conv_sp.h"


int launch_weight_images(const WeightImageGroup& g, int nblocks, hipStream_t st) {
  hipLaunchKernelGGL(sp_weight_image_kernel, dim3(nblocks), dim3(256), 0, st, g);
  return 0;
}
int launch_weight_image_table(const WeightImageTabEntry* tab, int n, int nblocks, hipStream_t st) {
  hipLaunchKernelGGL(sp_weight_image_table_kernel, dim3(nblocks), dim3(256), 0, st, tab, n);
  return 0;
}
int launch_ws_kernel(const IgemmArgs& a, int kind, int flip, int blocks, int ntotal, hipStream_t st, int ns) {
  const dim3 grid((unsigned)blocks);
#define WS1(NS_, K_, H_, N_, C_) if (ns == NS_ && kind == K_) { \
    if (flip) hipLaunchKernelGGL((igemm_patch_ws_kernel<NS_, H_, N_, C_, 1>), grid, dim3(512), 0, st, a, ntotal); \
    else hipLaunchKernelGGL((igemm_patch_ws_kernel<NS_, H_, N_, C_, 0>), grid, dim3(512), 0, st, a, ntotal); \
    return 0; }
  WS1(4, 1, 8, 3, 3) WS1(4, 2, 8, 6, 3) WS1(4, 3, 8, 4, 4) WS1(4, 4, 16, 3, 3)
  WS1(1, 1, 8, 3, 3) WS1(1, 2, 8, 6, 3) WS1(1, 3, 8, 4, 4) WS1(1, 4, 16, 3, 3)      // bf16: one piece, one product
#undef WS1
  return 1;
}
int launch_ws_group_kernel(const IgemmGroup& g, int flip, hipStream_t st, int ns) {
  const int end = g.blk_end[g.n - 1];
  if (ns == 1) {
    if (flip) hipLaunchKernelGGL((igemm_patch_ws_group_kernel<1, 1>), dim3(end), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((igemm_patch_ws_group_kernel<1, 0>), dim3(end), dim3(512), 0, st, g);
    return 0;
  }
  if (flip) hipLaunchKernelGGL((igemm_patch_ws_group_kernel<4, 1>), dim3(end), dim3(512), 0, st, g);
  else hipLaunchKernelGGL((igemm_patch_ws_group_kernel<4, 0>), dim3(end), dim3(512), 0, st, g);
  return 0;
}
