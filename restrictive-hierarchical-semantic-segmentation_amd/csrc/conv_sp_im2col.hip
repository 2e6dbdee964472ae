// Split-precision implicit GEMM, im2col body (1x1, stride 2, narrow images): kernel instances + launchers.
#define HRSEG_TU_IM2COL
#include "conv_common.h"
#include "conv_sp.h"


template <int NS>
static int launch_sp(const IgemmArgs& a, const SpPlan& pl, hipStream_t st) {
#define SP2(M_, N_) \
  if (pl.wtm == M_ && pl.wtn == N_) { \
    hipLaunchKernelGGL((igemm_sp_kernel<NS, M_, N_>), dim3(ceil_div(a.M, 64 * M_) * (a.N / (16 * N_)), pl.ksplit), dim3(256), 0, st, a); \
    return 0; }
#define SP1(M_) SP2(M_, 1) SP2(M_, 2) SP2(M_, 3) SP2(M_, 4) SP2(M_, 6)
  SP1(1) SP1(2) SP1(4)
#undef SP1
#undef SP2
  return 1;
}
int launch_sp_kernel(int ns, const IgemmArgs& a, const SpPlan& pl, hipStream_t st) {
  return ns == 4 ? launch_sp<4>(a, pl, st) : ns == 3 ? launch_sp<3>(a, pl, st) : ns == 2 ? launch_sp<2>(a, pl, st) : launch_sp<1>(a, pl, st);
}

template <int NS>
static int launch_sp_group(const IgemmGroup& g, int wtm, int wtn, bool full, hipStream_t st) {
  const dim3 grid(g.blk_end[g.n - 1]);
#define SPG(M_, N_) \
  if (wtm == M_ && wtn == N_) { \
    if (full) hipLaunchKernelGGL((igemm_sp_group_kernel<NS, M_, N_, true>), grid, dim3(256), 0, st, g); \
    else hipLaunchKernelGGL((igemm_sp_group_kernel<NS, M_, N_, false>), grid, dim3(256), 0, st, g); \
    return 0; }
  SPG(1, 3) SPG(1, 4) SPG(1, 6) SPG(2, 3) SPG(2, 4) SPG(2, 6)
#undef SPG
  return 1;
}
int launch_sp_group_kernel(int ns, const IgemmGroup& g, int wtm, int wtn, bool full, hipStream_t st) {
  return ns == 4 ? launch_sp_group<4>(g, wtm, wtn, full, st) : ns == 3 ? launch_sp_group<3>(g, wtm, wtn, full, st)
       : ns == 2 ? launch_sp_group<2>(g, wtm, wtn, full, st) : launch_sp_group<1>(g, wtm, wtn, full, st);
}

// wide channel tiles with pre-split weights (fp16x2): builds the weight image into `img`, then the convolution
int launch_spw_kernel(const IgemmArgs& a, int wtn, int ksplit, unsigned char* img, hipStream_t st) {
  const int BN = 16 * wtn;
  const int nslabs = (a.ntaps * (a.K / 16) + 1) / 2;
  hipLaunchKernelGGL(sp_weight_image_im2col_kernel, dim3((a.N / BN) * nslabs), dim3(256), 0, st, a.w, img, a.K, a.T, a.ntaps,
                     a.wtap_pk, a.wscale, BN, nslabs);
  IgemmArgs b = a;
  b.wimg = img;
  const dim3 grid(ceil_div(a.M, 128) * (a.N / BN), ksplit);
#define SPW(N_) if (wtn == N_) { hipLaunchKernelGGL((igemm_spw_kernel<2, N_>), grid, dim3(256), 0, st, b); return 0; }
  SPW(6) SPW(8) SPW(12) SPW(15)
#undef SPW
  return 1;
}
