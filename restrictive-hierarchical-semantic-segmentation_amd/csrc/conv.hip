// Convolution forward / data-gradient / weight-gradient for gfx950 (MI355X).
//
// Implicit GEMM on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32
// (64 lanes: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15]).
// Replaces nn.Conv2d of the reference (Models/models.py:113,116,322-324,
// 364-370,483-511,579-582,614,656,690,701) and its autograd backward.
//
//  * forward / dgrad: one kernel (igemm_conv).  GEMM rows are output pixels
//    gathered on the fly (im2col by address generation, nothing materialised
//    in HBM), columns are output channels; K runs over taps x 16-channel
//    chunks.  A = weights (M = channels), B = pixels (N), so a lane ends up
//    with 4 consecutive channels of one pixel -> one 16-byte NHWC store.
//    dgrad is the same kernel on dy with the transposed weights and a tap
//    table; stride-2 dgrad runs as 4 output-parity classes with 1/2/2/4 taps
//    (one grouped launch).  Up to 8 independent problems (the parallel HRNet
//    branches, the fuse paths of a module) share one launch (igemm_group).
//    Operands are staged with BUFFER loads (see make_rsrc).
//  * wgrad: dW[co][t][ci] = sum_pix dy[pix][co] * x[pix_t][ci]; each block owns
//    one (tap, cout tile, cin tile) and a pixel range, its 4 waves split the
//    pixels, reduce through LDS and add into dW with fp32 atomics (dW holds
//    the running gradient of the step, so the add IS the accumulation).
//  * Cin <= 8 (the image layer; with logit-concatenated re-encoding image + previous level's logits): direct VALU
//    kernels (conv_small.hip).
//
// This file holds the dispatch (tile plans, kernel-family choice per problem) and the C ABI; the kernel families
// and their launchers live in conv_f32.hip, conv_wgrad_f32.hip, conv_sp_im2col.hip, conv_sp_patch.hip,
// conv_sp_pgroup.hip, conv_ws.hip and conv_wgrad_sp.hip (split for build time; conv_common.h declares the launchers).
#include "conv_common.h"
#include "conv_sp.h"
#include <mutex>
#include <vector>

// launch counters per kernel family (hrseg_launch_count): the parity tests assert that a case really ran the family
// it claims to pin (e.g. the wave-specialised kernels on a 64x64 golden with lowered routing thresholds)
enum { CNT_WS = 0, CNT_WS_GROUP, CNT_PATCH_SP, CNT_SP_IM2COL, CNT_SP_PGROUP, CNT_SP_GROUP, CNT_F32, CNT_F32_GROUP, CNT_WGRAD_SP,
       CNT_WGRAD_F32, CNT_WGRAD_F32_GROUP, CNT_WGRAD9, CNT_SMALL_CIN, CNT_SP_WIDE, CNT_WS_CANVAS, CNT_WGRAD_SP_GROUP, CNT_WGRAD_SP_T5, CNT_WGRAD_SP_WIDE, CNT_N };
static const char* const g_cnt_names[CNT_N] = {"ws", "ws_group", "patch_sp", "sp_im2col", "sp_pgroup", "sp_group", "f32", "f32_group",
                                               "wgrad_sp", "wgrad_f32", "wgrad_f32_group", "wgrad9", "small_cin", "sp_wide", "ws_canvas", "wgrad_sp_group", "wgrad_sp_t5", "wgrad_sp_wide"};
static long g_cnt[CNT_N];
extern "C" long hrseg_launch_count(const char* family, int reset) {
  long total = 0;
  for (int i = 0; i < CNT_N; ++i)
    if (family ? !strcmp(family, g_cnt_names[i]) : (i != CNT_WS_CANVAS && i != CNT_WGRAD_SP_T5)) { total += g_cnt[i]; if (reset) g_cnt[i] = 0; }
  if (!family && reset) g_cnt[CNT_WS_CANVAS] = g_cnt[CNT_WGRAD_SP_T5] = 0;      // "ws_canvas" counts PROBLEMS laid out as a canvas inside ws / ws_group launches
  return total;
}

// tuning overrides (hrseg_tune, 0 = automatic): pixel tiles per wave, K chunks, LDS buffers, split-K
static int g_tune_wtm = 0, g_tune_kc = 0, g_tune_db = 0, g_tune_ksplit = 0;

// zero fill as a KERNEL node (not hipMemsetAsync): keeps a captured hipGraph a pure kernel chain
__global__ void zero_f32_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
static void zero_f32(float* p, size_t n, hipStream_t st) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
}

static IgemmPlan plan_igemm(const IgemmArgs& a) {
  IgemmPlan pl;
  // measured on MI355X (tools/conv_sweep.py): a single LDS buffer (more blocks per CU) beats double
  // buffering everywhere; large problems want 128-pixel tiles with 16-channel stages, small ones
  // 64-pixel tiles with the widest stage; under 256 blocks split K.
  pl.wtn = (a.N % 48 == 0) ? 3 : (a.N % 64 == 0) ? 4 : (a.N % 32 == 0) ? 2 : 1;
  const int kc_max = (a.K % 48 == 0) ? 3 : (a.K % 32 == 0) ? 2 : 1;
  const int ntn = a.N / (16 * pl.wtn);
  const long blocks128 = (long)ceil_div(a.M, 128) * ntn;
  const bool large = blocks128 >= 384;
  pl.wtm = large ? 2 : 1;
  pl.kc = large ? 1 : kc_max;
  pl.db = 1;
  pl.ksplit = 1;
  const long blocks = (long)ceil_div(a.M, 64 * pl.wtm) * ntn;
  const int nstages = a.ntaps * (a.K / (16 * pl.kc));
  if (blocks < 256) {
    // split only long reductions: under ~12 stages per slice the zero fill + atomics cost more than the
    // idle CUs (tools/conv_sweep.py: 48->384 3x3 s2 @39: 32 us split 3-way, 12 us unsplit)
    pl.ksplit = (int)((448 + blocks - 1) / blocks);
    if (pl.ksplit > nstages / 12) pl.ksplit = nstages / 12;
  }
  if (g_tune_wtm >= 10 && a.N % 96 == 0) pl.wtn = 6;   // tuning: tens digit 1 = 96-channel tiles
  if (g_tune_wtm % 10) pl.wtm = g_tune_wtm % 10;
  if (g_tune_kc && a.K % (16 * g_tune_kc) == 0) pl.kc = g_tune_kc;
  if (g_tune_db) pl.db = g_tune_db;
  if (g_tune_ksplit) pl.ksplit = g_tune_ksplit;
  if (hrseg_g_deterministic) pl.ksplit = 1;
  // split-K needs an output it may add into: accumulate mode, or a contiguous tensor it can zero
  const int nst = a.ntaps * (a.K / (16 * pl.kc));
  if (pl.ksplit > nst) pl.ksplit = nst;
  if (pl.ksplit < 1) pl.ksplit = 1;
  if (!(a.accumulate || (a.ldy == a.N && a.oys == 1 && a.oxs == 1))) pl.ksplit = 1;
  if (a.res || a.relu) pl.ksplit = 1;          // (a fused residual / ReLU epilogue needs the finished sum)
  return pl;
}

// derived fields + the host-side check behind the kernels' 32-bit buffer offsets: a tile (<= 256 GEMM
// rows) touches at most ceil(256 / (Ho*Wo)) + 1 consecutive images, which must span < 4 GB
static int finalize_args(IgemmArgs& a) {
  const bool big = (long)a.B * a.Ho * a.Wo >= (1L << 24);      // float-reciprocal division is exact below 2^24
  a.rcp_hw = big ? 0.f : 1.0f / (float)(a.Ho * a.Wo);
  a.rcp_w = big ? 0.f : 1.0f / (float)a.Wo;
  a.direct_out = (a.Hy == a.Ho && a.Wy == a.Wo && a.oys == 1 && a.oxs == 1 && a.oy0 == 0 && a.ox0 == 0) ? 1 : 0;
  a.wimg = nullptr;
  a.oy_min = a.ox_min = 0;
  for (int t = 0; t < a.ntaps; ++t) {
    const int oy = (int)((a.offy_pk >> (4 * t)) & 15) - 8, ox = (int)((a.offx_pk >> (4 * t)) & 15) - 8;
    if (oy < a.oy_min) a.oy_min = oy;
    if (ox < a.ox_min) a.ox_min = ox;
  }
  const double span = ((double)ceil_div(256, a.Ho * a.Wo) + 1.0) * a.Hi * a.Wi * (double)a.ldx * 4.0 +
                      (double)(2 * (-a.oy_min) * a.Wi + 2 * (-a.ox_min) + 16) * a.ldx * 4.0;
  const double wbytes = (double)a.N * a.T * a.K * 4.0;
  if (span >= 4294967296.0 || wbytes >= 4294967296.0) {
    hrseg_set_error("igemm: image of %dx%dx%d floats (or %g-byte weight) exceeds the 4 GB buffer-offset range",
                    a.Hi, a.Wi, a.ldx, wbytes);
    return HRSEG_ERR_UNSUPPORTED;
  }
  return 0;
}

// ---- split-precision plan: 128-pixel tiles for large problems, 64 otherwise; split-K under 256 blocks
static int g_sp_wtm = 0, g_sp_wtn = 0, g_sp_ksplit = 0;      // hrseg_tune overrides (0 = automatic)
static SpPlan plan_sp(const IgemmArgs& a) {
  SpPlan pl;
  pl.wtn = (a.N % 48 == 0) ? 3 : (a.N % 64 == 0) ? 4 : (a.N % 32 == 0) ? 2 : 1;
  if (g_sp_wtn && a.N % (16 * g_sp_wtn) == 0) pl.wtn = g_sp_wtn;
  const int ntn = a.N / (16 * pl.wtn);
  pl.wtm = ((long)ceil_div(a.M, 128) * ntn >= 512) ? 2 : 1;
  if (g_sp_wtm) pl.wtm = g_sp_wtm;
  const long blocks = (long)ceil_div(a.M, 64 * pl.wtm) * ntn;
  const int nslabs = (a.ntaps * (a.K / 16) + 1) / 2;
  pl.ksplit = 1;
  if (blocks < 256) {
    pl.ksplit = (int)((448 + blocks - 1) / blocks);
    if (pl.ksplit > nslabs / 8) pl.ksplit = nslabs / 8;
  }
  if (g_sp_ksplit) pl.ksplit = g_sp_ksplit;
  if (hrseg_g_deterministic) pl.ksplit = 1;
  if (pl.ksplit > nslabs) pl.ksplit = nslabs;
  if (pl.ksplit < 1) pl.ksplit = 1;
  if (!(a.accumulate || (a.ldy == a.N && a.oys == 1 && a.oxs == 1))) pl.ksplit = 1;
  if (a.res || a.relu) pl.ksplit = 1;          // (a fused residual / ReLU epilogue needs the finished sum)
  return pl;
}

// halo-patch body: full 3x3 stride-1 problems (forward or data gradient) on images wide enough that the
// 8 x 16 tiles waste little and fill the chip; returns the chunks per K stage (3 or 4), 0 = not a patch case
// routing thresholds (hrseg_tune keys sp_patch_min_tiles, auto_min_pixels, sp_ws_min_tiles): the parity tests lower them so that
// the small golden cases run the kernels the headline sizes run
static int g_patch_min_tiles = 192, g_auto_min_pix = 8192, g_ws_min_tiles = 96;
static int g_sp_patch = 1;              // hrseg_tune "sp_patch": 0 = never use the patch body
// tap geometry of a full 3x3 stride-1 problem: 0 = forward (tap t reads offset (t/3-1, t%3-1)), 1 = data gradient
// (offset (1-t/3, 1-t%3)), -1 = neither; the weight tap index must be t
static int patch_flip(const IgemmArgs& a) {
  if (a.ntaps != 9) return -1;
  bool fwd = true, bwd = true;
  for (int t = 0; t < 9; ++t) {
    const int oy = (int)((a.offy_pk >> (4 * t)) & 15) - 8, ox = (int)((a.offx_pk >> (4 * t)) & 15) - 8;
    if ((int)((a.wtap_pk >> (4 * t)) & 15) != t) return -1;
    fwd = fwd && oy == t / 3 - 1 && ox == t % 3 - 1;
    bwd = bwd && oy == 1 - t / 3 && ox == 1 - t % 3;
  }
  return fwd ? 0 : bwd ? 1 : -1;
}
static int patch_cs(const IgemmArgs& a, int wtn) {
  if (!g_sp_patch || patch_flip(a) < 0 || a.ntaps != 9 || a.T != 9 || a.sy != 1 || a.sx != 1 || !a.direct_out || a.Hi != a.Ho || a.Wi != a.Wo)
    return 0;
  if (a.oy_min != -1 || a.ox_min != -1 || (wtn != 3 && wtn != 4 && wtn != 6)) return 0;
  const int cs = (a.K % 48 == 0) ? 3 : (a.K % 64 == 0) ? 4 : 0;
  if (!cs) return 0;
  const long tiles = (long)a.B * ceil_div(a.Ho, 8) * ceil_div(a.Wo, 16);
  const double waste = (double)(ceil_div(a.Ho, 8) * 8) * (ceil_div(a.Wo, 16) * 16) / ((double)a.Ho * a.Wo);
  if (waste > 1.22 || tiles * (a.N / (16 * wtn)) < g_patch_min_tiles) return 0;
  return cs;
}
static long patch_tiles(const IgemmArgs& a, int wtn) {
  return (long)a.B * ceil_div(a.Ho, 8) * ceil_div(a.Wo, 16) * (a.N / (16 * wtn));
}
// ---- wave-specialised halo-patch path (fp16x2; conv_sp.h: igemm_patch_ws_body)
// The pre-split weight images live in a scratch buffer the host hands over once (hrseg_set_scratch; device memory is
// the caller's, as everywhere in this ABI).  It is cut into eight regions, one per stream that launches convolutions,
// each a ring: an image is written and read by kernels of ONE stream, in order, so reusing a slot after the ring
// wraps needs no synchronisation.  Without a scratch buffer the path is simply not taken.
extern int g_small_cin3;                // conv_small.hip
static int g_sp_ws = 1;                 // hrseg_tune "sp_ws": 0 = never use the wave-specialised body
static int g_ws_n48 = 1;                // hrseg_tune "sp_ws_n48": 0 = 48-channel tilings stay on the block-synchronous kernels
static int g_ws_waste = 200;            // hrseg_tune "sp_ws_waste": tile padding accepted, percent of the image
static int g_ws_bf16 = 1;               // hrseg_tune "sp_ws_bf16": 0 = the bf16 arithmetic (one piece, one product) stays off the wave-specialised kernels
static unsigned char* g_scratch = nullptr;
static size_t g_scratch_bytes = 0;
static int g_scratch_device = -1;       // the device that was current when the buffer was attached: launches on another one do not use it
static std::mutex g_scratch_mu;         // the region table is the only mutable state the launch path shares between host threads
struct ScratchRegion { hipStream_t st; bool used; size_t head; };
static const int SCRATCH_REGIONS = 8;
static ScratchRegion g_regions[SCRATCH_REGIONS];
extern "C" int hrseg_set_scratch(void* ptr, size_t bytes) {
  HRSEG_CHECK_ARG((ptr && bytes >= (1u << 20)) || (!ptr && bytes == 0), "hrseg_set_scratch: need a buffer of at least 1 MiB, or (null, 0)");
  HRSEG_CHECK_ARG(((uintptr_t)ptr & 255) == 0, "hrseg_set_scratch: the buffer must be 256-byte aligned");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  g_scratch = (unsigned char*)ptr;
  g_scratch_bytes = bytes;
  g_scratch_device = -1;
  if (ptr && hipGetDevice(&g_scratch_device) != hipSuccess) g_scratch_device = -1;
  for (auto& r : g_regions) r = ScratchRegion{nullptr, false, 0};
  return 0;
}
static bool scratch_usable() {
  if (!g_scratch) return false;
  int dev = -1;
  return hipGetDevice(&dev) == hipSuccess && dev == g_scratch_device;
}
// `bytes` CONTIGUOUS bytes of this stream's ring.  All images of one grouped launch are reserved together: they are written by one
// kernel and read by the next, so a wrap between two of them would put a later image over an earlier one of the same launch.
// nullptr: no buffer (or one of another device), no free region for a ninth stream, or more than a region holds.
static unsigned char* scratch_reserve(hipStream_t st, size_t bytes) {
  if (!scratch_usable()) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  const size_t region = (g_scratch_bytes / SCRATCH_REGIONS) & ~(size_t)255;
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > region) return nullptr;
  int r = -1;
  for (int i = 0; i < SCRATCH_REGIONS && r < 0; ++i)
    if (g_regions[i].used && g_regions[i].st == st) r = i;
  for (int i = 0; i < SCRATCH_REGIONS && r < 0; ++i)
    if (!g_regions[i].used) { g_regions[i] = ScratchRegion{st, true, 0}; r = i; }
  if (r < 0) return nullptr;
  if (g_regions[r].head + bytes > region) g_regions[r].head = 0;       // wrap BEFORE the group, never inside it
  unsigned char* p = g_scratch + (size_t)r * region + g_regions[r].head;
  g_regions[r].head += bytes;
  return p;
}
// tiling of the wave-specialised body for a problem: 0 = not eligible, 1 = 48 channels x 48-channel K stages on
// 8 x 16 pixel tiles, 2 = 96 x 48, 3 = 64 x 64, 4 = 48 x 48 on 16 x 16 pixel tiles (a 48-channel slab on 8 rows is
// 18 MFMAs per wave: too short for the producers to keep up)
static const int WS_WTN[5] = {0, 3, 6, 4, 3}, WS_CS[5] = {0, 3, 3, 4, 3}, WS_TH[5] = {0, 8, 8, 8, 16};
// Canvas mode of the wave-specialised body (conv_sp.h): the images of the batch side by side with a zero column between
// them, tiled as ONE image.  Taken when it cuts the padded area by at least 5 % (39 x 39: 1.26x -> 1.05x, 20 x 20: 1.92x ->
// 1.32x at 8 images; the 155 / 78-pixel branches stay per image) and the multiply-high image lookup is exact.
static int g_ws_canvas = 5;             // hrseg_tune "sp_ws_canvas": least cut of the padded area, percent (0 = per-image tiles everywhere)
static long ws_pixel_tiles(const IgemmArgs& a, int kind, bool canvas) {       // 16-column x TH-row tiles of the whole batch
  const long ty = ceil_div(a.Ho, WS_TH[kind]);
  return canvas ? ty * ceil_div((long)a.B * (a.Wo + 1) - 1, 16) : (long)a.B * ty * ceil_div(a.Wo, 16);
}
static bool ws_canvas(const IgemmArgs& a, int kind) {
  if (!g_ws_canvas || a.B < 2 || (long)a.B * (a.Wo + 1) >= 65536) return false;
  const long ldmax = a.ldx > a.ldy ? (a.ldx > a.ldr ? a.ldx : a.ldr) : (a.ldy > a.ldr ? a.ldy : a.ldr);
  if ((long)a.B * a.Ho * a.Wo * ldmax * 4 >= (1l << 31)) return false;          // one buffer descriptor (input, output, residual) spans the batch
  return ws_pixel_tiles(a, kind, true) * 100 <= ws_pixel_tiles(a, kind, false) * (100 - g_ws_canvas);
}
static long ws_tiles(const IgemmArgs& a, int kind) {
  return ws_pixel_tiles(a, kind, ws_canvas(a, kind)) * (a.N / (16 * WS_WTN[kind]));
}
static double ws_waste(const IgemmArgs& a, int kind) {
  return (double)ws_pixel_tiles(a, kind, ws_canvas(a, kind)) * WS_TH[kind] * 16 / ((double)a.B * a.Ho * a.Wo);
}
static void ws_set_canvas(IgemmArgs& a, int kind) {
  a.cv_w1 = a.cv_nb = 0;
  a.cv_magic = 0u;
  if (!ws_canvas(a, kind)) return;
  a.cv_w1 = a.Wo + 1;
  a.cv_nb = a.B;
  a.cv_magic = (unsigned)(((1ull << 32) + a.cv_w1 - 1) / a.cv_w1);
}
static int ws_kind(const IgemmArgs& a) {
  if (!g_sp_ws || !scratch_usable() || patch_flip(a) < 0 || a.T != 9 || a.sy != 1 || a.sx != 1 || !a.direct_out || a.Hi != a.Ho || a.Wi != a.Wo)
    return 0;
  if (a.oy_min != -1 || a.ox_min != -1) return 0;
  {   // 32-bit byte offsets inside one image (input patch, output, residual)
    const long ldmax = a.ldx > a.ldy ? (a.ldx > a.ldr ? a.ldx : a.ldr) : (a.ldy > a.ldr ? a.ldy : a.ldr);
    if ((double)a.Ho * a.Wo * (double)ldmax * 4.0 >= 4294967040.0) return 0;
  }
  int kind = (a.K % 48 == 0 && a.N % 48 == 0) ? 1 : (a.K % 64 == 0 && a.N % 64 == 0) ? 3 : 0;
  if (!kind) return 0;
  if (kind == 1) {      // the tiling is chosen on the per-image tile counts (a size class of the problem), canvas or not
    auto plain_tiles = [&](int k) { return ws_pixel_tiles(a, k, false) * (a.N / (16 * WS_WTN[k])); };
    if (a.N % 96 == 0 && plain_tiles(2) >= 160) kind = 2;
    else if (ws_pixel_tiles(a, 4, false) * 256 <= 1.10 * a.B * a.Ho * a.Wo && plain_tiles(4) >= 256) kind = 4;
  }
  if (ws_waste(a, kind) * 100 > g_ws_waste) return 0;
  if ((kind == 1 || kind == 4) && !g_ws_n48) return 0;
  return kind;
}
static size_t ws_image_bytes(const IgemmArgs& a, int kind, int ns) {      // ns = 4: two fp16 pieces per weight, 1: one bf16 piece
  const int wtn = WS_WTN[kind], cs = WS_CS[kind];
  return (size_t)(a.N / (16 * wtn)) * (a.K / (16 * cs)) * ((9 * cs + 1) / 2) * (size_t)((ns == 4 ? 2 : 1) * 16 * wtn * 64);
}
// ---- persistent weight images (hrseg_set_weight_image_arena / hrseg_weight_images_refresh) ------------------------------
// A weight image depends on the weights alone, and those change once per step: instead of one small image launch in front
// of every convolution (136 per HRNet step, 5.5 us + a kernel boundary each, all on the critical path) the images of the
// model's parameters live in an arena the caller owns and are rebuilt by ONE launch when the caller says the weights
// changed.  An image is cached only for a weight the caller flags as persistent (hrseg_conv_shape_t.w_persistent) AND that
// lies inside one of the two registered source ranges (the flat parameter buffer and its transposed copy): a scratch tensor
// that happens to reuse a dead model's addresses never hits.  First use of a weight registers it (and builds its image on
// the spot, as before); every later refresh rebuilds all registered images.  Single-threaded like the rest of the launch path.
struct ImgEntry { const float* w; unsigned char* img; int K, N, layout, ns, nblk; float wscale; };
static std::vector<ImgEntry> g_img;
static unsigned char* g_img_arena = nullptr;
static size_t g_img_arena_bytes = 0, g_img_arena_head = 0;
static WeightImageTabEntry* g_img_tab = nullptr;       // device copy of g_img for the refresh kernel (caller's memory)
static size_t g_img_tab_cap = 0;
static bool g_img_dirty = false;
static const float* g_img_range[4] = {nullptr, nullptr, nullptr, nullptr};
static int g_img_device = -1;
extern "C" int hrseg_set_weight_image_arena(void* arena, size_t bytes, void* table, size_t table_bytes, const float* lo0,
                                            const float* hi0, const float* lo1, const float* hi1) {
  HRSEG_CHECK_ARG((arena && bytes >= (1u << 20) && table && table_bytes >= sizeof(WeightImageTabEntry)) || (!arena && bytes == 0),
                  "hrseg_set_weight_image_arena: need an arena of at least 1 MiB and a table, or (null, 0)");
  HRSEG_CHECK_ARG((((uintptr_t)arena | (uintptr_t)table) & 255) == 0, "hrseg_set_weight_image_arena: buffers must be 256-byte aligned");
  g_img.clear();
  g_img_arena = (unsigned char*)arena;
  g_img_arena_bytes = bytes;
  g_img_arena_head = 0;
  g_img_tab = (WeightImageTabEntry*)table;
  g_img_tab_cap = arena ? table_bytes / sizeof(WeightImageTabEntry) : 0;
  g_img_dirty = false;
  g_img_range[0] = lo0; g_img_range[1] = hi0; g_img_range[2] = lo1; g_img_range[3] = hi1;
  g_img_device = -1;
  if (arena && hipGetDevice(&g_img_device) != hipSuccess) g_img_device = -1;
  return 0;
}
static bool img_cacheable(const IgemmArgs& a) {
  if (!g_img_arena || !a.w_persistent) return false;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != g_img_device) return false;
  return (a.w >= g_img_range[0] && a.w < g_img_range[1]) || (a.w >= g_img_range[2] && a.w < g_img_range[3]);
}
extern "C" int hrseg_weight_images_refresh(hrseg_stream_t stream) {
  if (g_img.empty()) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (g_img_dirty) {
    std::vector<WeightImageTabEntry> tab(g_img.size());
    int end = 0;
    for (size_t i = 0; i < g_img.size(); ++i) {
      end += g_img[i].nblk;
      tab[i] = WeightImageTabEntry{g_img[i].w, g_img[i].img, g_img[i].K, g_img[i].layout, g_img[i].ns, end, g_img[i].wscale, 0};
    }
    // (pageable source: the call returns once the table is staged; only after new weights were registered)
    if (hipMemcpyAsync(g_img_tab, tab.data(), tab.size() * sizeof(WeightImageTabEntry), hipMemcpyHostToDevice, st) != hipSuccess)
    { hrseg_set_error("hrseg_weight_images_refresh: table upload failed"); return HRSEG_ERR_LAUNCH; }
    g_img_dirty = false;
  }
  int total = 0;
  for (const auto& e : g_img) total += e.nblk;
  launch_weight_image_table(g_img_tab, (int)g_img.size(), total, st);
  HRSEG_LAUNCH_CHECK("weight_image_table");
  return 0;
}
// the weight images of n problems (sets a[i].wimg): cached ones are used as they are, the others are written with one launch
// (into their new arena slot, or into this stream's scratch ring); false: no scratch space
static bool ws_make_images(IgemmArgs* a, const int* kinds, int n, hipStream_t st, int ns) {
  WeightImageGroup g;
  g.n = 0;
  g.ns = ns;
  size_t off[MAXG], total = 0;
  int build[MAXG], nb = 0;
  unsigned char* dst[MAXG];
  for (int i = 0; i < n; ++i) {
    dst[i] = nullptr;
    const int layout = kinds[i] == 4 ? 1 : kinds[i];        // kinds 1 and 4 share the (48, 48) image layout
    const size_t bytes = (ws_image_bytes(a[i], kinds[i], ns) + 255) & ~(size_t)255;
    if (img_cacheable(a[i])) {
      for (const auto& e : g_img)
        if (e.w == a[i].w && e.layout == layout && e.ns == ns && e.K == a[i].K && e.N == a[i].N && e.wscale == a[i].wscale) { dst[i] = e.img; break; }
      if (dst[i]) continue;                                  // cached: kept current by hrseg_weight_images_refresh
      if (g_img_arena_head + bytes <= g_img_arena_bytes && g_img.size() < g_img_tab_cap) {
        const int wtn = WS_WTN[kinds[i]], cs = WS_CS[kinds[i]];
        dst[i] = g_img_arena + g_img_arena_head;
        g_img_arena_head += bytes;
        g_img.push_back(ImgEntry{a[i].w, dst[i], a[i].K, a[i].N, layout, ns, (a[i].N / (16 * wtn)) * (a[i].K / (16 * cs)) * ((9 * cs + 1) / 2),
                                 a[i].wscale});
        g_img_dirty = true;
        build[nb++] = i;
        continue;
      }
    }
    off[i] = total;
    total += bytes;
    build[nb++] = i;
  }
  unsigned char* base = total ? scratch_reserve(st, total) : nullptr;
  if (total && !base) return false;
  int end = 0;
  for (int j = 0; j < nb; ++j) {
    const int i = build[j];
    if (!dst[i]) dst[i] = base + off[i];
    const int wtn = WS_WTN[kinds[i]], cs = WS_CS[kinds[i]];
    end += (a[i].N / (16 * wtn)) * (a[i].K / (16 * cs)) * ((9 * cs + 1) / 2);
    g.blk_end[g.n] = end;
    g.kind[g.n] = kinds[i];
    g.K[g.n] = a[i].K;
    g.wscale[g.n] = a[i].wscale;
    g.w[g.n] = a[i].w;
    g.img[g.n] = dst[i];
    ++g.n;
  }
  for (int i = 0; i < n; ++i) a[i].wimg = dst[i];
  if (g.n) launch_weight_images(g, end, st);
  return true;
}
static int g_exp_nosplit = 0;           // hrseg_tune "exp_nosplit_x": MEASUREMENT ONLY -- the ceiling of "activations pre-split in HBM" (results wrong)
static int g_ws_epi_early = 1;          // hrseg_tune "ws_epi_early": 0 = the wave-specialised body reads accumulate / residual values at the tile's end
// hrseg_tune "ws_epi_cost" / "ws_epi_acc_cost": what a tile costs beyond its slabs, in slab times (0 = the defaults below,
// negative = none), for the block partition of a grouped launch.  A tile's prologue / epilogue (tile switch, 12 KB of stores per
// consumer wave, the reads of an accumulating or residual epilogue) is worth about ten slabs: a 48-channel layer's tile is ONE
// K stage of 14 slabs, so a partition by slab count alone starves it of blocks.  Measured in isolation (tools/ws_epilogue_ab.py,
// B = 8): four-branch forward 150 -> 121 us, three-branch 112 -> 91, two-branch 75.5 -> 67; accumulating data gradient
// 171 -> 126 / 127 -> 96 / 85 -> 75 us.  Results do not depend on the partition (no atomics, fixed per-tile order).
static int g_ws_epi_cost = 0, g_ws_epi_acc_cost = 0;
static const int WS_EPI_COST = 10, WS_EPI_ACC_COST = 16;
static int launch_ws_single(IgemmArgs a, int kind, hipStream_t st, int ns = 4) {
  if (a.x_presplit && ns != 4) return 1;                   // (the caller refuses the route: the pre-split form is fp16x2's)
  a.epi_early = g_ws_epi_early;
  a.exp_nosplit = (g_exp_nosplit && !patch_flip(a)) ? 1 : 0;      // (forward only: the data gradient's operand is a scaled gradient)
  ws_set_canvas(a, kind);
  const int ntotal = (int)ws_tiles(a, kind);
  if (!ws_make_images(&a, &kind, 1, st, ns)) return 1;
  const int per = ceil_div(ntotal, 256);
  const dim3 grid((unsigned)ceil_div(ntotal, per));
  const int flip = patch_flip(a);
  if (a.stat_partial) *a.stat_rows = (int)grid.x;           // one row of partial sums per block
  ++g_cnt[CNT_WS];
  g_cnt[CNT_WS_CANVAS] += a.cv_w1 > 0;
  return launch_ws_kernel(a, kind, flip, (int)grid.x, ntotal, st, ns);
}
// One launch for several problems: the 256 persistent blocks are divided among the problems in proportion to their
// slab counts, then blocks move from the problem that finishes first to the one that finishes last while that helps.
static int launch_ws_group(IgemmArgs* a, const int* kinds, int n, hipStream_t st, int ns = 4) {
  IgemmGroup g;
  g.n = n;
  long cost[MAXG], ntot[MAXG], total = 0;
  int blocks[MAXG], flip = -1;
  for (int i = 0; i < n; ++i) {
    if (a[i].x_presplit && ns != 4) return 1;
    const int f = patch_flip(a[i]);
    if (flip >= 0 && f != flip) return 1;
    flip = f;
    const int cs = WS_CS[kinds[i]];
    ws_set_canvas(a[i], kinds[i]);
    ntot[i] = ws_tiles(a[i], kinds[i]);
    a[i].epi_early = g_ws_epi_early;
    a[i].exp_nosplit = (g_exp_nosplit && !f) ? 1 : 0;
    // slabs per tile, plus the tile's epilogue in slab times (an accumulating / residual epilogue waits for its reads)
    const int ec = g_ws_epi_cost ? (g_ws_epi_cost > 0 ? g_ws_epi_cost : 0) : WS_EPI_COST;
    const int eac = g_ws_epi_acc_cost ? (g_ws_epi_acc_cost > 0 ? g_ws_epi_acc_cost : 0) : WS_EPI_ACC_COST;
    cost[i] = (long)(a[i].K / (16 * cs)) * ((9 * cs + 1) / 2) + ((a[i].accumulate || a[i].res) ? eac : ec);
    total += ntot[i] * cost[i];
  }
  int used = 0;
  for (int i = 0; i < n; ++i) {
    blocks[i] = (int)(256 * ntot[i] * cost[i] / total);
    if (blocks[i] < 1) blocks[i] = 1;
    if (blocks[i] > ntot[i]) blocks[i] = (int)ntot[i];
    used += blocks[i];
  }
  auto span = [&](int i, int b) { return (long)ceil_div((int)ntot[i], b) * cost[i]; };
  for (int it = 0; it < 512; ++it) {
    int hi = 0, lo = -1;
    for (int i = 1; i < n; ++i)
      if (span(i, blocks[i]) > span(hi, blocks[hi])) hi = i;
    if (blocks[hi] >= ntot[hi]) break;
    if (used < 256) { ++blocks[hi]; ++used; continue; }
    for (int i = 0; i < n; ++i)          // the donor: the problem that stays shortest with one block less
      if (i != hi && blocks[i] > 1 && (lo < 0 || span(i, blocks[i] - 1) < span(lo, blocks[lo] - 1))) lo = i;
    if (lo < 0 || span(lo, blocks[lo] - 1) >= span(hi, blocks[hi])) break;
    --blocks[lo];
    ++blocks[hi];
  }
  if (!ws_make_images(a, kinds, n, st, ns)) return 1;
  int end = 0;
  for (int i = 0; i < n; ++i) {
    const int per = ceil_div((int)ntot[i], blocks[i]);
    g.tiles[i] = ceil_div((int)ntot[i], per);      // blocks that have work
    if (a[i].stat_partial) *a[i].stat_rows = g.tiles[i];
    g.ksplit[i] = (int)ntot[i];
    end += g.tiles[i];
    g.blk_end[i] = end;
    g.kind[i] = kinds[i];
    g.a[i] = a[i];
    g_cnt[CNT_WS_CANVAS] += a[i].cv_w1 > 0;
  }
  ++g_cnt[CNT_WS_GROUP];
  return launch_ws_group_kernel(g, flip, st, ns);
}

static int g_sp_img = 1;               // hrseg_tune "sp_img": 0 = the block-synchronous patch body splits its weights on the fly
static int g_sp_persist = 2;            // hrseg_tune "sp_persist": persistent patch blocks per CU (0 = one tile per block)
static int launch_patch_sp(int ns, const IgemmArgs& a_in, int wtn, int cs, hipStream_t st) {
  const int ntotal = (int)patch_tiles(a_in, wtn);
  int blocks = ntotal;
  if (g_sp_persist > 0 && ntotal > 256 * g_sp_persist) {
    // equal chunks: the block count that gives every block the same number of tiles (+-1)
    const int per = ceil_div(ntotal, 256 * g_sp_persist);
    blocks = ceil_div(ntotal, per);
  }
  const int flip = patch_flip(a_in);
  IgemmArgs a = a_in;
  if (ns == 4 && g_sp_img) {      // pre-split weights where an image layout exists for the tiling (else on the fly)
    int kind = (wtn == 3 && cs == 3) ? 1 : (wtn == 6 && cs == 3) ? 2 : (wtn == 4 && cs == 4) ? 3 : 0;
    if (kind) ws_make_images(&a, &kind, 1, st, 4);      // (no scratch space: a.wimg stays null)
  }
  ++g_cnt[CNT_PATCH_SP];
  return launch_patch_sp_kernel(ns, a, wtn, cs, flip, blocks, ntotal, st);
}

// wide-tile im2col body (conv_sp.h: igemm_spw_body): channel tile in 16-channel units, 0 = not a case for it.
// At least 96 output channels, a tile count that fills the chip (128-pixel tiles), and room in the scratch ring.
static int g_sp_wide = 1;               // hrseg_tune "sp_wide": 0 = never use the wide-tile body
static int g_spw_min_blocks = 256;      // hrseg_tune "sp_wide_min_blocks"
static int spw_wtn(const IgemmArgs& a) {
  if (!g_sp_wide || !scratch_usable() || a.K % 16 || a.M < 128) return 0;
  const int wtn = (a.N % 240 == 0) ? 15 : (a.N % 192 == 0) ? 12 : (a.N % 128 == 0) ? 8 : (a.N % 96 == 0) ? 6 : 0;
  if (!wtn) return 0;
  if ((long)ceil_div(a.M, 128) * (a.N / (16 * wtn)) < g_spw_min_blocks) return 0;
  // the body walks its slabs six at a time (register sets and LDS buffers are compile-time): short reductions would
  // multiply zeros for most of a trip -- they stay on the narrow body (hrseg_tune sp_wide = 2 lifts the rule: tests)
  const int nslabs = (a.ntaps * (a.K / 16) + 1) / 2, padded = (nslabs + 5) / 6 * 6;
  if (g_sp_wide < 2 && (padded - nslabs) * 10 > nslabs) return 0;
  return wtn;
}

static int launch_sp(int ns, const IgemmArgs& a, const SpPlan& pl, hipStream_t st) {
  if (launch_sp_kernel(ns, a, pl, st) == 0) { ++g_cnt[CNT_SP_IM2COL]; return 0; }
  hrseg_set_error("igemm_sp: no kernel for plan wtm=%d wtn=%d", pl.wtm, pl.wtn);
  return HRSEG_ERR_UNSUPPORTED;
}

static int sp_pieces(int precision) {    // hrseg_conv_precision -> split scheme of conv_sp.h (0: the fp32 MFMA kernels)
  return precision == HRSEG_CONV_BF16X3 ? 3 : precision == HRSEG_CONV_BF16X2 ? 2 : precision == HRSEG_CONV_BF16 ? 1
       : precision == HRSEG_CONV_FP16X2 ? 4 : 0;      // AUTO is resolved per problem before this is asked
}
// fp16x2: weights are scaled by a fixed 2^8 before the split (|w| up to 255 stays in fp16 range, typical
// |w| ~ 0.01-1 keeps its low piece out of the subnormals)
static bool is_f16(int precision) { return precision == HRSEG_CONV_FP16X2 || precision == HRSEG_CONV_AUTO; }
static void set_sp_scales(IgemmArgs& a, int precision, const float* grad_absmax) {
  const bool f16 = is_f16(precision);
  a.wscale = f16 ? 256.f : 1.f;
  a.wscale_inv = f16 ? 1.f / 256.f : 1.f;
  a.xmax = f16 ? grad_absmax : nullptr;
}

// HRSEG_CONV_AUTO: fp32-grade results from whichever kernel family is faster for the problem (measured on
// MI355X, tools/sp_bench.py): forward / data gradient: the fp16x2 kernels for problems of at least 8192 output
// pixels (the halo-patch body on wide 3x3 stride-1 images, the im2col body otherwise), the fp32 MFMA kernels for
// the small ones; weight gradients: fp16x2 throughout
static int resolve_auto(const IgemmArgs& a, int precision) {
  if (precision != HRSEG_CONV_AUTO) return precision;
  if (a.M >= g_auto_min_pix) return HRSEG_CONV_FP16X2;
  const int k = ws_kind(a);           // the wave-specialised patch body also wins on small images
  return (k && ws_tiles(a, k) >= g_ws_min_tiles) ? HRSEG_CONV_FP16X2 : HRSEG_CONV_F32;
}

// a pre-split pixel operand (hrseg_conv_shape_t.x_split) is readable by the wave-specialised forward kernels and the nine-tap
// weight gradient only: every other route refuses it loudly instead of reading the bytes as fp32
static int x_split_unsupported(const char* who) {
  hrseg_set_error("%s: x_split is set but this problem does not run the wave-specialised / nine-tap kernels (ask hrseg_conv_x_split_ok first)", who);
  return HRSEG_ERR_UNSUPPORTED;
}
static int dispatch_igemm(const IgemmArgs& a_in, int precision, hipStream_t st) {
  IgemmArgs a = a_in;
  if (int e = finalize_args(a)) return e;
  precision = resolve_auto(a, precision);
  if (const int ns = sp_pieces(precision)) {
    const SpPlan pl = plan_sp(a);
    if ((ns == 4 || (ns == 1 && g_ws_bf16)) && !g_sp_wtn) {
      const int kind = ws_kind(a);
      if (kind && ws_tiles(a, kind) >= g_ws_min_tiles && launch_ws_single(a, kind, st, ns) == 0) return 0;
    }
    if (a.x_presplit) return x_split_unsupported("hrseg_conv_fwd");
    if (const int cs = patch_cs(a, pl.wtn)) {
      const int rc = launch_patch_sp(ns, a, pl.wtn, cs, st);
      if (rc == 0) return 0;
    }
    if (ns == 4 && !g_sp_wtn) {
      // wide channel tiles + pre-split weights (igemm_spw_body): layers of at least 96 output channels on large images
      const int wtn = spw_wtn(a);
      if (wtn) {
        const size_t bytes = (size_t)(a.N / (16 * wtn)) * ((a.ntaps * (a.K / 16) + 1) / 2) * (size_t)(2 * 16 * wtn * 64);
        if (unsigned char* img = scratch_reserve(st, bytes)) {
          ++g_cnt[CNT_SP_WIDE];
          if (launch_spw_kernel(a, wtn, 1, img, st) == 0) return 0;
        }
      }
    }
    if (pl.ksplit > 1 && !a.accumulate) zero_f32(a.y, (size_t)a.B * a.Hy * a.Wy * a.N, st);
    return launch_sp(ns, a, pl, st);
  }
  if (a.x_presplit) return x_split_unsupported("hrseg_conv_fwd");
  IgemmPlan pl = plan_igemm(a);
  if (pl.ksplit > 1 && !a.accumulate) zero_f32(a.y, (size_t)a.B * a.Hy * a.Wy * a.N, st);
  if (launch_igemm_f32(a, pl, st) == 0) { ++g_cnt[CNT_F32]; return 0; }
  hrseg_set_error("igemm: no kernel for plan wtm=%d wtn=%d kc=%d db=%d", pl.wtm, pl.wtn, pl.kc, pl.db);
  return HRSEG_ERR_UNSUPPORTED;
}

// group dispatch: one common plan (64-pixel tiles, widest K stage, single LDS buffer); problems whose
// channel tiling differs from the first one's, or that need a zero-fill they cannot get, make the
// caller fall back to per-problem launches (return 1).
static int g_group_wtm = 0;     // tuning override of the grouped launches' pixel tile (0 = automatic, 1 = 64, 2 = 128 pixels)

static int launch_sp_group(int ns, const IgemmGroup& g_in, int wtm, int wtn, int cs, hipStream_t st) {
  IgemmGroup g = g_in;
  if (ns == 4 && cs && g_sp_img && ((wtn == 3 && cs == 3) || (wtn == 4 && cs == 4))) {
    IgemmArgs im[MAXG];
    int kinds[MAXG], idx[MAXG], m = 0;
    for (int i = 0; i < g.n; ++i)
      if (g.kind[i]) { im[m] = g.a[i]; kinds[m] = wtn == 3 ? 1 : 3; idx[m++] = i; }
    if (m && ws_make_images(im, kinds, m, st, 4))
      for (int j = 0; j < m; ++j) g.a[idx[j]].wimg = im[j].wimg;
  }
  bool full = true;
  for (int i = 0; i < g.n; ++i) full = full && g.a[i].ntaps == 9 && g.a[i].T == 9 && g.a[i].sy == 1 && g.a[i].oys == 1;
  if (cs) {      // at least one problem runs the halo-patch body (all of them with the same tap geometry)
    int flip = -1;
    for (int i = 0; i < g.n; ++i)
      if (g.kind[i]) {
        const int f = patch_flip(g.a[i]);
        if (flip >= 0 && f != flip) return 1;
        flip = f;
      }
    ++g_cnt[CNT_SP_PGROUP];
    return launch_sp_pgroup_kernel(ns, g, wtm, wtn, cs, flip, st);
  }
  ++g_cnt[CNT_SP_GROUP];
  return launch_sp_group_kernel(ns, g, wtm, wtn, full, st);
}

static int dispatch_igemm_group(const IgemmArgs* a, int n, int precision, hipStream_t st);
// HRSEG_CONV_AUTO on a group: the problems the halo-patch body takes go out as one fp16x2 launch (a
// low-resolution branch inside a patch launch would run at the patch body's occupancy); of the rest, the small ones
// (< 8192 pixels) as one fp32 launch and the others one by one
// Contract of the group dispatchers: 1 = nothing was launched (the caller issues the problems one by one), 0 = all launched,
// negative = hard error after some launches went out (the caller must NOT re-issue: accumulating problems would add twice)
static int dispatch_igemm_group_auto(const IgemmArgs* a, int n, hipStream_t st, bool try_ws = true) {
  if (try_ws && g_sp_ws && scratch_usable() && !g_sp_wtn) {
    // the problems the wave-specialised body takes go out as one launch of 256 persistent blocks; the rest as before
    IgemmArgs wsa[MAXG], rest[MAXG];
    int kinds[MAXG], nw = 0, nr = 0;
    for (int i = 0; i < n; ++i) {
      IgemmArgs f = a[i];
      if (finalize_args(f)) return 1;
      const int k = ws_kind(f);
      if (k && ws_tiles(f, k) >= g_ws_min_tiles) { wsa[nw] = f; kinds[nw++] = k; }
      else rest[nr++] = a[i];
    }
    // (a lone taker next to a 48-channel high-resolution branch: the two share one halo-patch group launch instead,
    // measured 79 us against 41 + 49)
    bool pair = false;
    if (nw == 1 && nr == 1) {
      IgemmArgs f = rest[0];
      pair = finalize_args(f) == 0 && f.N % 48 == 0 && patch_cs(f, 3) == 3;
    }
    bool split_rest = false;
    for (int i = 0; i < nr; ++i) split_rest = split_rest || rest[i].x_presplit;
    if (split_rest || (pair && (wsa[0].x_presplit))) return x_split_unsupported("hrseg_conv_fwd_group");
    if (nw >= 1 && !pair && launch_ws_group(wsa, kinds, nw, st) == 0) {
      if (nr == 0) return 0;
      if (nr == 1) { const int e = dispatch_igemm(rest[0], HRSEG_CONV_AUTO, st); return e < 0 ? e : (e ? HRSEG_ERR_LAUNCH : 0); }
      if (dispatch_igemm_group_auto(rest, nr, st, false) == 0) return 0;
      for (int i = 0; i < nr; ++i)
        if (int e = dispatch_igemm(rest[i], HRSEG_CONV_AUTO, st)) return e < 0 ? e : HRSEG_ERR_LAUNCH;
      return 0;
    }
  }
  for (int i = 0; i < n; ++i)
    if (a[i].x_presplit) return x_split_unsupported("hrseg_conv_fwd_group");
  IgemmArgs hi[MAXG], lo[MAXG];
  int nh = 0, nl = 0;
  for (int i = 0; i < n; ++i) {
    IgemmArgs f = a[i];
    if (finalize_args(f)) return 1;
    const int wtn = (f.N % 48 == 0) ? 3 : (f.N % 64 == 0) ? 4 : 0;
    if (wtn && patch_cs(f, wtn) == wtn) hi[nh++] = a[i];
    else lo[nl++] = a[i];
  }
  if (nh == 0) {
    bool small = true, large = true;
    for (int i = 0; i < n; ++i) { small = small && a[i].M < g_auto_min_pix; large = large && a[i].M >= g_auto_min_pix; }
    if (small) return dispatch_igemm_group(a, n, HRSEG_CONV_F32, st);
    if (large) return dispatch_igemm_group(a, n, HRSEG_CONV_FP16X2, st);
  }
  int rc = (nh >= 2) ? dispatch_igemm_group(hi, nh, HRSEG_CONV_FP16X2, st) : 1;
  if (rc != 0)
    for (int i = 0; i < nh; ++i)
      if (int e = dispatch_igemm(hi[i], HRSEG_CONV_FP16X2, st)) return e < 0 ? e : HRSEG_ERR_LAUNCH;
  IgemmArgs sm[MAXG];
  int nsm = 0;
  for (int i = 0; i < nl; ++i) {
    if (lo[i].M < g_auto_min_pix) { sm[nsm++] = lo[i]; continue; }
    if (int e = dispatch_igemm(lo[i], HRSEG_CONV_FP16X2, st)) return e < 0 ? e : HRSEG_ERR_LAUNCH;
  }
  rc = (nsm >= 2) ? dispatch_igemm_group(sm, nsm, HRSEG_CONV_F32, st) : 1;
  if (rc != 0)
    for (int i = 0; i < nsm; ++i)
      if (int e = dispatch_igemm(sm[i], HRSEG_CONV_F32, st)) return e < 0 ? e : HRSEG_ERR_LAUNCH;
  return 0;
}

static int dispatch_igemm_group(const IgemmArgs* a, int n, int precision, hipStream_t st) {
  if (n < 2 || n > MAXG || g_tune_wtm || g_tune_kc || g_tune_db || g_tune_ksplit) return 1;
  if (precision == HRSEG_CONV_AUTO) return dispatch_igemm_group_auto(a, n, st);
  for (int i = 0; i < n; ++i)
    if (a[i].x_presplit) return x_split_unsupported("hrseg_conv_fwd_group");
  const int ns = sp_pieces(precision);
  if (ns == 1 && g_ws_bf16 && g_sp_ws && scratch_usable() && !g_sp_wtn) {
    // bf16 arithmetic (BASELINE configs[4]): the problems the wave-specialised body takes go out as one launch of its
    // one-piece instance, the rest as before
    IgemmArgs wsa[MAXG], rest[MAXG];
    int kinds[MAXG], nw = 0, nr = 0;
    for (int i = 0; i < n; ++i) {
      IgemmArgs f = a[i];
      if (finalize_args(f)) return 1;
      const int k = ws_kind(f);
      if (k && ws_tiles(f, k) >= g_ws_min_tiles) { wsa[nw] = f; kinds[nw++] = k; }
      else rest[nr++] = a[i];
    }
    if (nw >= 1 && launch_ws_group(wsa, kinds, nw, st, 1) == 0) {
      for (int i = 0; i < nr; ++i)
        if (int e = dispatch_igemm(rest[i], precision, st)) return e < 0 ? e : HRSEG_ERR_LAUNCH;
      return 0;
    }
  }
  int wtn = (a[0].N % 48 == 0) ? 3 : (a[0].N % 64 == 0) ? 4 : 0;
  const int kc = ns ? 1 : (a[0].K % 48 == 0) ? 3 : (a[0].K % 32 == 0) ? 2 : 1;
  if (!wtn) return 1;
  if (ns && g_sp_wtn) {
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && a[i].N % (16 * g_sp_wtn) == 0;
    if (ok && (g_sp_wtn == 3 || g_sp_wtn == 4 || g_sp_wtn == 6)) wtn = g_sp_wtn;
  }
  IgemmGroup g;
  g.n = n;
  for (int i = 0; i < n; ++i)
    if (a[i].N % (16 * wtn) || a[i].K % (16 * kc)) return 1;
  // per-problem split-K, then order the problems by stages per block, longest first: the blocks
  // that run longest must not be the ones dispatched last (the grid's tail)
  int tiles[MAXG], ks[MAXG], work[MAXG], order[MAXG], kind[MAXG];
  const int wtm = ns ? (g_sp_wtm ? g_sp_wtm : 2) : (g_group_wtm ? g_group_wtm : 1);
  if (ns && wtm > 2) return 1;
  int group_cs = 0;
  IgemmArgs fa[MAXG];
  for (int i = 0; i < n; ++i) {
    fa[i] = a[i];
    if (finalize_args(fa[i])) return 1;    // per-problem launches report the error
    kind[i] = 0;
    if (ns && (wtn == 3 || wtn == 4)) {
      const int cs = patch_cs(fa[i], wtn);
      if (cs && cs == wtn && (!group_cs || group_cs == cs)) { kind[i] = 1; group_cs = cs; }   // instances: (wtn, cs) = (3,3), (4,4)
    }
  }
  for (int i = 0; i < n; ++i) {
    tiles[i] = kind[i] ? (int)patch_tiles(fa[i], wtn) : ceil_div(a[i].M, 64 * wtm) * (a[i].N / (16 * wtn));
    const int nstages = ns ? (a[i].ntaps * (a[i].K / 16) + 1) / 2 : a[i].ntaps * (a[i].K / (16 * kc));
    ks[i] = 1;
    const bool can_split = !kind[i] && !a[i].res && !a[i].relu &&
                           (a[i].accumulate || (a[i].ldy == a[i].N && a[i].oys == 1 && a[i].oxs == 1));
    if (can_split && tiles[i] < 512 && !hrseg_g_deterministic) {
      ks[i] = ceil_div(512, tiles[i]);
      const int min_stages = ns ? 8 : 12;        // stages (fp32: 16*kc channels; split precision: 32-channel slabs) per slice
      if (ks[i] > nstages / min_stages) ks[i] = nstages / min_stages;
      if (ks[i] < 1) ks[i] = 1;
    }
    if (ks[i] > 1 && !a[i].accumulate) zero_f32(a[i].y, (size_t)a[i].B * a[i].Hy * a[i].Wy * a[i].N, st);
    work[i] = ceil_div(nstages, ks[i]);
    order[i] = i;
  }
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (work[order[j]] > work[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  int end = 0;
  for (int o = 0; o < n; ++o) {
    const int i = order[o];
    g.tiles[o] = tiles[i];
    g.ksplit[o] = ks[i];
    end += tiles[i] * ks[i];
    g.blk_end[o] = end;
    g.kind[o] = kind[i];
    g.a[o] = fa[i];
  }
  if (ns) return launch_sp_group(ns, g, wtm, wtn, group_cs, st);
  ++g_cnt[CNT_F32_GROUP];
  return launch_igemm_group_f32(g, wtm, wtn, kc, st);
}

static int g_tune_wg_pix = 0, g_tune_wg_db = 0, g_tune_wg_blocks = 0;

// host-side check behind the 32-bit buffer offsets of wgrad_body: one block's pixel range
int check_wgrad_span(const WgradArgs& a) {
  const double imgs = (double)ceil_div(a.pix_per_block, a.Ho * a.Wo) + 1.0;
  if (imgs * a.Hi * a.Wi * (double)a.ldx * 4.0 >= 4294967296.0 || (double)a.pix_per_block * a.lddy * 4.0 >= 4294967296.0) {
    hrseg_set_error("wgrad: a block's pixel range (%d pixels) exceeds the 4 GB buffer-offset range", a.pix_per_block);
    return HRSEG_ERR_UNSUPPORTED;
  }
  return 0;
}

static int g_wg_t5 = 1;                // hrseg_tune "wgrad_sp_t5": 0 = no 80 x 80 tiles in the tap-per-block weight gradient
static int g_wg_wide = 1;              // hrseg_tune "wgrad_sp_wide": 0 = never the wide-tile weight-gradient body
// pixel ranges per tile set such that the blocks fill whole rounds of `slots` (one block per CU): the k in [kmin, kmax] with the
// best fill, the smallest such k on ties
static int ksplit_for_rounds(int tiles, int kmin, int kmax, int slots) {
  int best = 0;
  double best_eff = 0.0;
  for (int k = kmin; k <= kmax; ++k) {
    const int blocks = tiles * k;
    const double eff = (double)blocks / (ceil_div(blocks, slots) * (double)slots);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = k; }
  }
  return best;
}
static int dispatch_wgrad_sp(int ns, WgradArgs a, hipStream_t st) {
  // wide layers whose channels divide into 240 x 144 block tiles (the 720 -> 720 head layer): wgrad_spw_body
  if (ns == 4 && g_wg_wide && a.Cout % 240 == 0 && a.Cin % 144 == 0 && a.M >= 65536 && !hrseg_g_deterministic && !g_tune_wg_blocks) {
    const int tiles = (a.Cout / 240) * (a.Cin / 144) * a.T;
    int kmax = a.M / 2048;                       // at least 64 stages of 32 pixels per block
    if (kmax > 64) kmax = 64;
    const int ksplit = ksplit_for_rounds(tiles, 1, kmax, 256);
    a.pix_per_block = ceil_div(ceil_div(a.M, ksplit), 32) * 32;
    if (int e = check_wgrad_span(a)) return e;
    ++g_cnt[CNT_WGRAD_SP_WIDE];
    return launch_wgrad_spw_kernel(a, ceil_div(a.M, a.pix_per_block), tiles, st);
  }
  int tn = (a.Cout % 48 == 0) ? 3 : (a.Cout % 64 == 0) ? 4 : (a.Cout % 32 == 0) ? 2 : 1;
  int tk = (a.Cin % 48 == 0) ? 3 : (a.Cin % 64 == 0) ? 4 : (a.Cin % 32 == 0) ? 2 : 1;
  constexpr int PIX = 128;                      // SpWgradLds::PIX
  // Wide layers (720 -> 720 of the HRNet head): 80 x 80 tiles.  The body splits both operands on the fly, (tn + tk) * 12 VALU
  // per wave and 32 pixels next to tn * tk * 3 MFMAs: at 48 x 48 the split outweighs the MFMAs (MFMA-busy 0.19) and every
  // operand row is pulled Cin / 48 resp. Cout / 48 = 15 times through L2; at 80 x 80 it is 120 VALU : 75 MFMAs and 9 pulls.
  // One block per CU (80 KB of LDS): the pixel ranges are sized so that the blocks fill whole rounds of 256.
  const bool t5 = ns == 4 && g_wg_t5 && a.Cout % 80 == 0 && a.Cin % 80 == 0 && (long)a.Cout * a.Cin >= 240 * 240 && a.M >= 65536;
  if (t5) tn = tk = 5;
  const int tiles = (a.Cout / (16 * tn)) * (a.Cin / (16 * tk)) * a.T;
  int target = 7 * tiles;
  if (target < 512) target = 512;
  if (target > 4096) target = 4096;
  if (t5) {
    int kmax = a.M / (PIX * 8);
    if (kmax > 32) kmax = 32;
    if (const int best = ksplit_for_rounds(tiles, 4, kmax, 256)) target = best * tiles;
  }
  if (g_tune_wg_blocks) target = g_tune_wg_blocks;
  int ksplit = target / tiles;
  if (ksplit < 1 || hrseg_g_deterministic) ksplit = 1;
  int ppb = ceil_div(ceil_div(a.M, ksplit), PIX) * PIX;
  if (ppb < 2 * PIX) ppb = 2 * PIX;
  a.pix_per_block = ppb;
  if (int e = check_wgrad_span(a)) return e;
  const int gx = ceil_div(a.M, ppb);
  ++g_cnt[CNT_WGRAD_SP];
  g_cnt[CNT_WGRAD_SP_T5] += t5;      // ("wgrad_sp" launches that used 80 x 80 tiles)
  return launch_wgrad_sp_kernel(ns, a, tn, tk, gx, tiles, st);
}

static int g_wg_mult = 0, g_wg_min = 0, g_wg_max = 0;      // tuning overrides of the grouped weight-gradient grid
static void plan_wgrad_blocks(WgradArgs& a, int tn, int tk, int pix, int& gx, int& tiles) {
  tiles = (a.Cout / (16 * tn)) * (a.Cin / (16 * tk)) * a.T;
  // measured (tools/wgrad_group_plan.py, groups of 2-4 branch convs at 4 and 8 images): 2 pixel ranges per tile
  // set, at least 768 and at most 2048 blocks per problem -- 7 ranges per tile set left the 384-channel
  // problem with 4032 blocks of 8 stages whose cross-wave reduction and atomics cost as much as their MFMAs
  int target = (g_wg_mult ? g_wg_mult : 2) * tiles;
  const int tmin = g_wg_min ? g_wg_min : 768, tmax = g_wg_max ? g_wg_max : 2048;
  if (target < tmin) target = tmin;
  if (target > tmax) target = tmax;
  int ksplit = target / tiles;
  if (ksplit < 1 || hrseg_g_deterministic) ksplit = 1;
  int ppb = ceil_div(ceil_div(a.M, ksplit), pix) * pix;
  if (ppb < 4 * pix) ppb = 4 * pix;
  a.pix_per_block = ppb;
  gx = ceil_div(a.M, ppb);
}

// returns 1 when the problems cannot share one kernel instance (caller falls back)
static int g_wg_group_sp = 1;          // hrseg_tune "wgrad_group_sp": 0 = grouped tap-per-block weight gradients stay on the fp32 kernel
static int dispatch_wgrad_group(WgradArgs* a, int n, int prec, hipStream_t st) {
  if (n < 2 || n > MAXG || g_tune_wg_pix || g_tune_wg_db || g_tune_wg_blocks) return 1;
  const bool sp = prec == HRSEG_CONV_FP16X2 && g_wg_group_sp;       // every problem asked for fp16x2 (AUTO resolves to it)
  const int tn = (a[0].Cout % 48 == 0) ? 3 : (a[0].Cout % 64 == 0) ? 4 : 0;
  const int tk = (a[0].Cin % 48 == 0) ? 3 : (a[0].Cin % 64 == 0) ? 4 : 0;
  if (!tn || !tk) return 1;
  WgradGroup g;
  g.n = n;
  int end = 0;
  for (int i = 0; i < n; ++i) {
    if (a[i].Cout % (16 * tn) || a[i].Cin % (16 * tk)) return 1;
    int gx, tiles;
    plan_wgrad_blocks(a[i], tn, tk, sp ? 128 : 64, gx, tiles);     // (pixels per stage: SpWgradLds::PIX resp. the fp32 plan)
    if (check_wgrad_span(a[i])) return 1;   // the per-problem launch reports the error
    g.gx[i] = gx;
    end += gx * tiles;
    g.blk_end[i] = end;
    g.a[i] = a[i];
  }
  if (sp) {
    ++g_cnt[CNT_WGRAD_SP_GROUP];
    return launch_wgrad_group_sp(g, tn, tk, end, st);
  }
  ++g_cnt[CNT_WGRAD_F32_GROUP];
  return launch_wgrad_group_f32(g, tn, tk, end, st);
}

// --------------------------------------------------------------------------- nine-tap weight gradient (workspace + ordered reduce)
static int g_wg9_blocks = 0;           // hrseg_tune "wgrad9_blocks": target blocks per problem (0 = the table below)
static int g_wg9_blocks_n[WG9_MAXG + 1] = {0, 0, 0, 0, 0};   // hrseg_tune "wgrad9_blocks1" .. "wgrad9_blocks4": the same, per group size
static int g_wg9 = 1;                  // hrseg_tune "wgrad9": 0 = never use the nine-tap kernel
// tiles per side of the dW tile (3: channels multiple of 48, 4: multiple of 64), 0 = not a nine-tap case
static int wgrad9_tnk(const hrseg_conv_shape_t& s) {
  if (!g_wg9 || s.ksize != 3 || s.stride != 1 || sp_pieces(s.precision) == 0) return 0;
  if (s.Cin % 48 == 0 && s.Cout % 48 == 0) return 3;
  if (s.Cin % 64 == 0 && s.Cout % 64 == 0) return 4;
  return 0;
}
// Target blocks per problem by the number of problems sharing the launch.  Alone on the GPU the kernel likes whole rounds
// of 512 resident blocks (256 per problem; tools/wgrad9_sweep.py), but in the train step it runs on the side stream beside
// the data-gradient chain, and there fewer, longer blocks win: less workspace to write and reduce, and CUs left over for
// the main stream.  Step-level sweep at the headline geometry (bench.py, 20 steps, two runs each, profiles/README.md r04):
// 2 problems 128 (-0.2 ms against 256), 3 problems 128 (-0.65 ms against 160), 4 problems 96 (-0.5 ms against 256);
// a single problem (UNet, HRNet stage 1) keeps 256 -- UNet loses 1.5 ms at 128.
static const int WG9_TARGET[5] = {256, 256, 128, 128, 96};
static void wgrad9_plan(const hrseg_conv_shape_t& s, int tnk, Wgrad9Args& a, int group_n) {
  a.B = s.B; a.H = s.Hi; a.W = s.Wi; a.Cin = s.Cin; a.Cout = s.Cout; a.ldx = s.ldx; a.lddy = s.ldy;
  a.tiles_x = ceil_div(s.Wi, 16); a.tiles_y = ceil_div(s.Hi, 4);
  a.ntiles = s.B * a.tiles_x * a.tiles_y;
  const int npairs = (s.Cout / (16 * tnk)) * (s.Cin / (16 * tnk));
  const int target = g_wg9_blocks_n[group_n] ? g_wg9_blocks_n[group_n] : g_wg9_blocks ? g_wg9_blocks : WG9_TARGET[group_n < 4 ? group_n : 4];
  int chunks = ceil_div(target, npairs);
  if (chunks > a.ntiles) chunks = a.ntiles;
  if (chunks < 1) chunks = 1;
  a.per = ceil_div(a.ntiles, chunks);
  a.nchunks = ceil_div(a.ntiles, a.per);
}
// plan of one call: every problem's pixel chunks (all problems share one tiling and one arithmetic)
struct Wg9Plan { int tnk; Wgrad9Args a[WG9_MAXG]; };
static bool wgrad9_plan_all(int n, const hrseg_conv_shape_t* shapes, Wg9Plan& pl) {
  if (n < 1 || n > WG9_MAXG) return false;
  pl.tnk = wgrad9_tnk(shapes[0]);
  if (!pl.tnk) return false;
  for (int i = 0; i < n; ++i) {
    if (wgrad9_tnk(shapes[i]) != pl.tnk || shapes[i].precision != shapes[0].precision) return false;
    wgrad9_plan(shapes[i], pl.tnk, pl.a[i], n);
  }
  return true;
}
// bytes of workspace the nine-tap path wants for these problems (0: not applicable to all of them)
static size_t wgrad9_ws_bytes(int n, const hrseg_conv_shape_t* shapes) {
  Wg9Plan pl;
  if (!wgrad9_plan_all(n, shapes, pl)) return 0;
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += (size_t)pl.a[i].nchunks * shapes[i].Cout * 9 * shapes[i].Cin * 4;
  return total;
}
static int dispatch_wgrad9(int n, const float* const* x, const float* const* dy, float* const* dw,
                           const hrseg_conv_shape_t* shapes, float* ws, hipStream_t st) {
  const int ns = sp_pieces(shapes[0].precision);
  for (int i = 0; i < n; ++i)
    if (shapes[i].x_split && ns != 4) return x_split_unsupported("hrseg_conv_wgrad_group_ws");
  Wg9Plan pl;
  if (!wgrad9_plan_all(n, shapes, pl)) return HRSEG_ERR_UNSUPPORTED;      // (the caller asked wgrad9_ws_bytes first)
  const int tnk = pl.tnk;
  Wgrad9Group g;
  Wgrad9Reduce r;
  g.n = r.n = n;
  int end = 0, rend = 0;
  for (int i = 0; i < n; ++i) {
    Wgrad9Args& a = pl.a[i];
    a.x = x[i]; a.dy = dy[i]; a.ws = ws;
    a.exp_nosplit = g_exp_nosplit;
    a.x_presplit = shapes[i].x_split;
    a.dymax = shapes[i].precision == HRSEG_CONV_FP16X2 ? shapes[i].grad_absmax : nullptr;
    HRSEG_CHECK_ARG((double)shapes[i].Hi * shapes[i].Wi * (double)(shapes[i].ldx > shapes[i].ldy ? shapes[i].ldx : shapes[i].ldy) * 4.0 < 4294967296.0,
                    "wgrad9: one image exceeds the 4 GB buffer-offset range");
    const long elems = (long)shapes[i].Cout * 9 * shapes[i].Cin;
    end += (shapes[i].Cout / (16 * tnk)) * (shapes[i].Cin / (16 * tnk)) * a.nchunks;
    g.blk_end[i] = end;
    g.a[i] = a;
    r.ws[i] = ws; r.dw[i] = dw[i]; r.nchunks[i] = a.nchunks; r.n4[i] = elems / 4;
    rend += (int)((elems / 4 + 31) / 32 < 2048 ? (elems / 4 + 31) / 32 : 2048);
    r.blk_end[i] = rend;
    ws += (size_t)a.nchunks * elems;
  }
  ++g_cnt[CNT_WGRAD9];
  return launch_wgrad9_kernels(ns, tnk, g, end, r, rend, st);
}

// --------------------------------------------------------------------------- weight transpose
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int T,
                                        int Cin) {
  // wt[ci][t][co] = w[co][t][ci]; tile 32x32 through LDS, one tap per blockIdx.z
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int co = co0 + j, ci = ci0 + threadIdx.x;
    tile[j][threadIdx.x] = (co < Cout && ci < Cin) ? w[((size_t)co * T + t) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int ci = ci0 + j, co = co0 + threadIdx.x;
    if (co < Cout && ci < Cin) wt[((size_t)ci * T + t) * Cout + co] = tile[threadIdx.x][j];
  }
}

// every conv weight of the flat parameter buffer in ONE launch: entry e = one 32x32 (co,ci) tile of
// one tap of one weight; {slot offset, Cout, T, Cin, co0, ci0, tap, -}
__global__ void weight_transpose_all_kernel(const float* __restrict__ flat, float* __restrict__ flat_t,
                                            const int* __restrict__ table) {
  __shared__ float tile[32][33];
  const int* e = table + (size_t)blockIdx.x * 8;
  const int off = e[0], Cout = e[1], T = e[2], Cin = e[3], co0 = e[4], ci0 = e[5], t = e[6];
  const float* w = flat + off;
  float* wt = flat_t + off;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int co = co0 + j, ci = ci0 + threadIdx.x;
    tile[j][threadIdx.x] = (co < Cout && ci < Cin) ? w[((size_t)co * T + t) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int ci = ci0 + j, co = co0 + threadIdx.x;
    if (co < Cout && ci < Cin) wt[((size_t)ci * T + t) * Cout + co] = tile[threadIdx.x][j];
  }
}

// --------------------------------------------------------------------------- C ABI
static int check_shape(const hrseg_conv_shape_t* s, const char* who) {
  HRSEG_CHECK_ARG(s != nullptr, "%s: null shape", who);
  HRSEG_CHECK_ARG(s->ksize == 1 || s->ksize == 3, "%s: ksize %d not in {1,3}", who, s->ksize);
  HRSEG_CHECK_ARG(s->precision >= HRSEG_CONV_F32 && s->precision <= HRSEG_CONV_FP16X2, "%s: precision %d is not a hrseg_conv_precision",
                  who, s->precision);
  HRSEG_CHECK_ARG(s->stride == 1 || s->stride == 2, "%s: stride %d not in {1,2}", who, s->stride);
  HRSEG_CHECK_ARG(s->B > 0 && s->Hi > 0 && s->Wi > 0 && s->Cin > 0 && s->Cout > 0, "%s: non-positive dims", who);
  const int pad = (s->ksize - 1) / 2;
  const int ho = (s->Hi + 2 * pad - s->ksize) / s->stride + 1, wo = (s->Wi + 2 * pad - s->ksize) / s->stride + 1;
  HRSEG_CHECK_ARG(ho == s->Ho && wo == s->Wo, "%s: output %dx%d does not match input %dx%d k%d s%d (expect %dx%d)",
                  who, s->Ho, s->Wo, s->Hi, s->Wi, s->ksize, s->stride, ho, wo);
  HRSEG_CHECK_ARG(s->ldx >= s->Cin && s->ldy >= s->Cout, "%s: ld smaller than channel count", who);
  HRSEG_CHECK_ARG(!s->residual || (s->ldr >= s->Cout && s->ldr % 4 == 0), "%s: residual stride %d (Cout %d)", who, s->ldr, s->Cout);
  HRSEG_CHECK_ARG((long)s->B * s->Hi * s->Wi < (1L << 31) && (long)s->B * s->Ho * s->Wo < (1L << 31),
                  "%s: more than 2^31 pixels per tensor is not supported (32-bit pixel indices)", who);
  return 0;
}

static void pack_taps(IgemmArgs& a, int n, const int* oy, const int* ox, const int* wt) {
  a.ntaps = n;
  a.offy_pk = a.offx_pk = a.wtap_pk = 0;
  for (int t = 0; t < n; ++t) {
    a.offy_pk |= (unsigned long long)(oy[t] + 8) << (4 * t);
    a.offx_pk |= (unsigned long long)(ox[t] + 8) << (4 * t);
    a.wtap_pk |= (unsigned long long)wt[t] << (4 * t);
  }
}

static void fill_fwd_args(IgemmArgs& a, const float* x, const float* w, const float* bias, float* y,
                          const hrseg_conv_shape_t* s) {
  a = IgemmArgs{};
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.ldx = s->ldx; a.ldy = s->ldy;
  a.B = s->B; a.Hi = s->Hi; a.Wi = s->Wi; a.K = s->Cin;
  a.Ho = s->Ho; a.Wo = s->Wo; a.N = s->Cout; a.M = s->B * s->Ho * s->Wo;
  a.sy = a.sx = s->stride;
  a.Hy = s->Ho; a.Wy = s->Wo; a.oys = a.oxs = 1; a.oy0 = a.ox0 = 0;
  a.T = s->ksize * s->ksize; a.accumulate = 0;
  int oy[9], ox[9], wt[9];
  const int pad = (s->ksize - 1) / 2;
  for (int t = 0; t < a.T; ++t) { oy[t] = t / s->ksize - pad; ox[t] = t % s->ksize - pad; wt[t] = t; }
  pack_taps(a, a.T, oy, ox, wt);
  set_sp_scales(a, s->precision, nullptr);
  a.res = s->residual; a.ldr = s->ldr; a.relu = s->relu;
  a.w_persistent = s->w_persistent;
  a.x_presplit = s->x_split;
  // BatchNorm statistics in the epilogue: offered to the launchers only where the caller gave both pointers, the output is the
  // BatchNorm's input as stored (no fused residual / ReLU) and the run need not be bit-reproducible
  if (s->stat_partial && s->stat_rows) {
    *s->stat_rows = 0;
    if (!hrseg_g_deterministic && !s->residual && !s->relu && s->Cout <= 1024) { a.stat_partial = s->stat_partial; a.stat_rows = s->stat_rows; }
  }
}

// stride-1 data gradient as a forward-style gather over dy with the transposed weights
static void fill_dgrad_s1_args(IgemmArgs& a, const float* dy, const float* wt, float* dx, int accumulate,
                               const hrseg_conv_shape_t* s) {
  a = IgemmArgs{};
  a.x = dy; a.w = wt; a.bias = nullptr; a.y = dx; a.ldx = s->ldy; a.ldy = s->ldx;
  a.B = s->B; a.Hi = s->Ho; a.Wi = s->Wo; a.K = s->Cout;
  a.N = s->Cin; a.T = s->ksize * s->ksize; a.accumulate = accumulate;
  a.Hy = s->Hi; a.Wy = s->Wi;
  a.Ho = s->Hi; a.Wo = s->Wi; a.M = s->B * s->Hi * s->Wi;
  a.sy = a.sx = 1; a.oys = a.oxs = 1; a.oy0 = a.ox0 = 0;
  const int ks = s->ksize, pad = (ks - 1) / 2;
  int oy[9], ox[9], wtp[9];
  for (int t = 0; t < a.T; ++t) { oy[t] = pad - t / ks; ox[t] = pad - t % ks; wtp[t] = t; }
  pack_taps(a, a.T, oy, ox, wtp);
  set_sp_scales(a, s->precision, s->grad_absmax);
  a.w_persistent = s->w_persistent;
}

static void fill_wgrad_args(WgradArgs& a, const float* x, const float* dy, float* dw, const hrseg_conv_shape_t* s) {
  a = WgradArgs{};
  a.x = x; a.dy = dy; a.dw = dw; a.ldx = s->ldx; a.lddy = s->ldy;
  a.B = s->B; a.Hi = s->Hi; a.Wi = s->Wi; a.Cin = s->Cin; a.Ho = s->Ho; a.Wo = s->Wo; a.Cout = s->Cout;
  a.M = s->B * s->Ho * s->Wo; a.ks = s->ksize; a.stride = s->stride; a.T = s->ksize * s->ksize;
  const bool big = (long)s->B * s->Ho * s->Wo >= (1L << 24);   // float-reciprocal division is exact below 2^24
  a.rcp_hw = big ? 0.f : 1.0f / (float)(s->Ho * s->Wo);
  a.rcp_w = big ? 0.f : 1.0f / (float)s->Wo;
  a.dymax = s->precision == HRSEG_CONV_FP16X2 ? s->grad_absmax : nullptr;
}

static bool mfma_shape(const hrseg_conv_shape_t* s) {
  return s->Cin % 16 == 0 && s->Cout % 16 == 0 && s->ldx % 4 == 0 && s->ldy % 4 == 0;
}

// weight gradients under AUTO run fp16x2 (faster than the fp32 kernels on every measured shape)
static void resolve_wgrad_shapes(int n, const hrseg_conv_shape_t* in, hrseg_conv_shape_t* out) {
  for (int i = 0; i < n; ++i) {
    out[i] = in[i];
    if (out[i].precision == HRSEG_CONV_AUTO) out[i].precision = HRSEG_CONV_FP16X2;
  }
}

// ---- grouped entry points: n independent problems, one launch when they can share a kernel
extern "C" int hrseg_conv_fwd_group(int n, const float* const* x, const float* const* w, const float* const* bias,
                                    float* const* y, const hrseg_conv_shape_t* shapes, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(n >= 1 && x && w && y && shapes, "hrseg_conv_fwd_group: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  bool ok = n <= MAXG;
  for (int i = 0; i < n; ++i) {
    if (int e = check_shape(&shapes[i], "hrseg_conv_fwd_group")) return e;
    ok = ok && mfma_shape(&shapes[i]) && shapes[i].precision == shapes[0].precision;
  }
  if (ok && n >= 2) {
    IgemmArgs a[MAXG];
    for (int i = 0; i < n; ++i) fill_fwd_args(a[i], x[i], w[i], bias ? bias[i] : nullptr, y[i], &shapes[i]);
    const int rc = dispatch_igemm_group(a, n, shapes[0].precision, st);
    if (rc < 0) return rc;
    if (rc == 0) {
      HRSEG_LAUNCH_CHECK("igemm_group(fwd)");
      return 0;
    }
  }
  for (int i = 0; i < n; ++i)
    if (int e = hrseg_conv_fwd(x[i], w[i], bias ? bias[i] : nullptr, y[i], &shapes[i], stream)) return e;
  return 0;
}

extern "C" int hrseg_conv_dgrad_group(int n, const float* const* dy, const float* const* wt, float* const* dx,
                                      const int* accumulate, const hrseg_conv_shape_t* shapes,
                                      hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(n >= 1 && dy && wt && dx && accumulate && shapes, "hrseg_conv_dgrad_group: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  bool ok = n <= MAXG;
  for (int i = 0; i < n; ++i) {
    if (int e = check_shape(&shapes[i], "hrseg_conv_dgrad_group")) return e;
    ok = ok && mfma_shape(&shapes[i]) && shapes[i].stride == 1 && shapes[i].precision == shapes[0].precision;
  }
  if (ok && n >= 2) {
    IgemmArgs a[MAXG];
    for (int i = 0; i < n; ++i) fill_dgrad_s1_args(a[i], dy[i], wt[i], dx[i], accumulate[i], &shapes[i]);
    const int rc = dispatch_igemm_group(a, n, shapes[0].precision, st);
    if (rc < 0) return rc;
    if (rc == 0) {
      HRSEG_LAUNCH_CHECK("igemm_group(dgrad)");
      return 0;
    }
  }
  for (int i = 0; i < n; ++i)
    if (int e = hrseg_conv_dgrad(dy[i], wt[i], dx[i], accumulate[i], &shapes[i], stream)) return e;
  return 0;
}

extern "C" int hrseg_conv_wgrad_group(int n, const float* const* x, const float* const* dy, float* const* dw,
                                      const hrseg_conv_shape_t* shapes, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(n >= 1 && x && dy && dw && shapes, "hrseg_conv_wgrad_group: bad arguments");
  for (int i = 0; i < n; ++i)
    if (shapes[i].x_split) return x_split_unsupported("hrseg_conv_wgrad_group");
  hipStream_t st = (hipStream_t)stream;
  bool ok = n <= MAXG;
  for (int i = 0; i < n; ++i) {
    if (int e = check_shape(&shapes[i], "hrseg_conv_wgrad_group")) return e;
    ok = ok && mfma_shape(&shapes[i]);
  }
  if (ok && n >= 2) {
    WgradArgs a[MAXG];
    hrseg_conv_shape_t rs[MAXG];
    resolve_wgrad_shapes(n, shapes, rs);
    int prec = rs[0].precision;
    for (int i = 0; i < n; ++i) {
      fill_wgrad_args(a[i], x[i], dy[i], dw[i], &rs[i]);
      if (rs[i].precision != prec) prec = HRSEG_CONV_F32;
    }
    if (dispatch_wgrad_group(a, n, prec, st) == 0) {
      HRSEG_LAUNCH_CHECK("wgrad_group");
      return 0;
    }
  }
  for (int i = 0; i < n; ++i)
    if (int e = hrseg_conv_wgrad(x[i], dy[i], dw[i], &shapes[i], stream)) return e;
  return 0;
}

extern "C" size_t hrseg_conv_wgrad_workspace_bytes(int n, const hrseg_conv_shape_t* shapes_in) {
  if (!shapes_in || n < 1 || n > WG9_MAXG) return 0;
  hrseg_conv_shape_t shapes[WG9_MAXG];
  resolve_wgrad_shapes(n, shapes_in, shapes);
  for (int i = 0; i < n; ++i)
    if (check_shape(&shapes[i], "hrseg_conv_wgrad_workspace_bytes")) return 0;
  return wgrad9_ws_bytes(n, shapes);
}

static int g_x_split = 1;              // hrseg_tune "x_split": 0 = hrseg_conv_x_split_ok always answers no (activations stay fp32 everywhere)
// Mirrors the routing of hrseg_conv_fwd(_group) and hrseg_conv_wgrad_group_ws for these shapes: yes only when EVERY problem of
// the forward call goes out through the wave-specialised kernels (fp16x2 arithmetic) and the weight gradients through the
// nine-tap kernel.  The calls themselves refuse x_split on any other route, so a mismatch is an error, never a misread tensor.
extern "C" int hrseg_conv_x_split_ok(int n, const hrseg_conv_shape_t* shapes) {
  if (!g_x_split || !shapes || n < 1 || n > MAXG || n > WG9_MAXG) return 0;
  if (!g_sp_ws || !scratch_usable() || g_sp_wtn || g_tune_wtm || g_tune_kc || g_tune_db || g_tune_ksplit) return 0;
  for (int i = 0; i < n; ++i) {
    const hrseg_conv_shape_t& s = shapes[i];
    if (s.ksize != 3 || s.stride != 1 || s.residual || s.relu || !mfma_shape(&s) || s.precision != shapes[0].precision) return 0;
    if (s.precision != HRSEG_CONV_AUTO && !(s.precision == HRSEG_CONV_FP16X2 && n == 1)) return 0;      // (explicit fp16x2 GROUPS run the block-synchronous kernels)
    if (s.B < 1 || s.Hi < 1 || s.Wi < 1 || s.Ho != s.Hi || s.Wo != s.Wi) return 0;
    IgemmArgs a;
    fill_fwd_args(a, reinterpret_cast<const float*>(256), reinterpret_cast<const float*>(256), nullptr, reinterpret_cast<float*>(256), &s);
    a.stat_partial = nullptr; a.stat_rows = nullptr;
    if (finalize_args(a)) return 0;
    const int k = ws_kind(a);
    if (!k || ws_tiles(a, k) < g_ws_min_tiles) return 0;
  }
  hrseg_conv_shape_t rs[WG9_MAXG];
  resolve_wgrad_shapes(n, shapes, rs);
  for (int i = 0; i < n; ++i)
    if (sp_pieces(rs[i].precision) != 4) return 0;
  return wgrad9_ws_bytes(n, rs) > 0 ? 1 : 0;
}

extern "C" int hrseg_conv_wgrad_group_ws(int n, const float* const* x, const float* const* dy, float* const* dw,
                                         const hrseg_conv_shape_t* shapes_in, void* workspace, size_t workspace_bytes,
                                         hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(n >= 1 && x && dy && dw && shapes_in, "hrseg_conv_wgrad_group_ws: bad arguments");
  if (n > WG9_MAXG) return hrseg_conv_wgrad_group(n, x, dy, dw, shapes_in, stream);
  hrseg_conv_shape_t shapes[WG9_MAXG];
  resolve_wgrad_shapes(n, shapes_in, shapes);
  for (int i = 0; i < n; ++i)
    if (int e = check_shape(&shapes[i], "hrseg_conv_wgrad_group_ws")) return e;
  const size_t need = wgrad9_ws_bytes(n, shapes);
  if (need && workspace && workspace_bytes >= need) {
    for (int i = 0; i < n; ++i) HRSEG_CHECK_ARG(x[i] && dy[i] && dw[i], "hrseg_conv_wgrad_group_ws: null pointer");
    return dispatch_wgrad9(n, x, dy, dw, shapes, (float*)workspace, (hipStream_t)stream);
  }
  return hrseg_conv_wgrad_group(n, x, dy, dw, shapes, stream);
}

extern "C" int hrseg_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                              const hrseg_conv_shape_t* s, hrseg_stream_t stream) {
  if (int e = check_shape(s, "hrseg_conv_fwd")) return e;
  HRSEG_CHECK_ARG(x && w && y, "hrseg_conv_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  HRSEG_CHECK_ARG(s->Cout % 16 == 0 || s->Cin <= HRSEG_SMALL_CIN_MAX, "hrseg_conv_fwd: Cout %d not a multiple of 16", s->Cout);
  if (s->Cin <= HRSEG_SMALL_CIN_MAX) {
    launch_small_cin_fwd(x, w, bias, y, s, st);
    ++g_cnt[CNT_SMALL_CIN];
    HRSEG_LAUNCH_CHECK("conv_small_cin_fwd");
    return 0;
  }
  HRSEG_CHECK_ARG(s->Cin % 16 == 0, "hrseg_conv_fwd: Cin %d not a multiple of 16", s->Cin);
  HRSEG_CHECK_ARG(s->ldx % 4 == 0 && s->ldy % 4 == 0, "hrseg_conv_fwd: ld must be a multiple of 4");
  IgemmArgs a;
  fill_fwd_args(a, x, w, bias, y, s);
  if (int e = dispatch_igemm(a, s->precision, st)) return e;
  HRSEG_LAUNCH_CHECK("igemm_conv(fwd)");
  return 0;
}

extern "C" int hrseg_conv_dgrad(const float* dy, const float* wt, float* dx, int accumulate,
                                const hrseg_conv_shape_t* s, hrseg_stream_t stream) {
  if (int e = check_shape(s, "hrseg_conv_dgrad")) return e;
  HRSEG_CHECK_ARG(dy && wt && dx, "hrseg_conv_dgrad: null pointer");
  if (s->Cin <= HRSEG_SMALL_CIN_MAX) {
    // first layer with a differentiable input (logit-concatenated re-encoding): `wt` is the FORWARD weight [Cout][k*k][Cin]
    launch_small_cin_dgrad(dy, wt, dx, accumulate, s, (hipStream_t)stream);
    ++g_cnt[CNT_SMALL_CIN];
    HRSEG_LAUNCH_CHECK("conv_small_cin_dgrad");
    return 0;
  }
  HRSEG_CHECK_ARG(s->Cin % 16 == 0 && s->Cout % 16 == 0, "hrseg_conv_dgrad: channels (%d,%d) must be multiples of 16",
                  s->Cin, s->Cout);
  HRSEG_CHECK_ARG(s->ldx % 4 == 0 && s->ldy % 4 == 0, "hrseg_conv_dgrad: ld must be a multiple of 4");
  hipStream_t st = (hipStream_t)stream;
  IgemmArgs a{};
  a.x = dy; a.w = wt; a.bias = nullptr; a.y = dx; a.ldx = s->ldy; a.ldy = s->ldx;
  a.B = s->B; a.Hi = s->Ho; a.Wi = s->Wo; a.K = s->Cout;  // GEMM input = dy
  a.N = s->Cin; a.T = s->ksize * s->ksize; a.accumulate = accumulate;
  a.Hy = s->Hi; a.Wy = s->Wi;
  set_sp_scales(a, s->precision, s->grad_absmax);
  const int ks = s->ksize, pad = (ks - 1) / 2;
  if (s->stride == 1) {
    a.Ho = s->Hi; a.Wo = s->Wi; a.M = s->B * s->Hi * s->Wi;
    a.sy = a.sx = 1; a.oys = a.oxs = 1; a.oy0 = a.ox0 = 0;
    int oy[9], ox[9], wtp[9];
    for (int t = 0; t < a.T; ++t) { oy[t] = pad - t / ks; ox[t] = pad - t % ks; wtp[t] = t; }
    pack_taps(a, a.T, oy, ox, wtp);
    if (int e = dispatch_igemm(a, s->precision, st)) return e;
    HRSEG_LAUNCH_CHECK("igemm_conv(dgrad)");
    return 0;
  }
  // stride 2: dx[y,x] = sum_{kh,kw : (y+pad-kh) even, (x+pad-kw) even} dy[(y+pad-kh)/2, (x+pad-kw)/2] w[kh,kw]
  // The four output-parity classes write disjoint pixels of dx: one grouped launch.
  IgemmArgs cls[4];
  int ncls = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int hc = (s->Hi - py + 1) / 2, wc = (s->Wi - px + 1) / 2;  // pixels of this parity class
      if (hc <= 0 || wc <= 0) continue;
      int oy[9], ox[9], wtp[9], n = 0;
      for (int kh = 0; kh < ks; ++kh)
        for (int kw = 0; kw < ks; ++kw) {
          if (((py + pad - kh) & 1) || ((px + pad - kw) & 1)) continue;
          // y = 2*yy+py  ->  dy row = yy + (py+pad-kh)/2 (arithmetic shift handles -1)
          oy[n] = (py + pad - kh) >> 1;
          ox[n] = (px + pad - kw) >> 1;
          wtp[n] = kh * ks + kw;
          ++n;
        }
      IgemmArgs c = a;
      c.Ho = hc; c.Wo = wc; c.M = s->B * hc * wc;
      c.sy = c.sx = 1; c.oys = c.oxs = 2; c.oy0 = py; c.ox0 = px;
      pack_taps(c, n, oy, ox, wtp);
      cls[ncls++] = c;
    }
  if (ncls >= 2) {
    const int rc = dispatch_igemm_group(cls, ncls, s->precision, st);
    if (rc < 0) return rc;
    if (rc == 0) {
      HRSEG_LAUNCH_CHECK("igemm_group(dgrad s2)");
      return 0;
    }
  }
  for (int i = 0; i < ncls; ++i) {
    if (int e = dispatch_igemm(cls[i], s->precision, st)) return e;
    HRSEG_LAUNCH_CHECK("igemm_conv(dgrad s2)");
  }
  return 0;
}

extern "C" int hrseg_conv_wgrad(const float* x, const float* dy, float* dw, const hrseg_conv_shape_t* s_in,
                                hrseg_stream_t stream) {
  if (s_in && s_in->x_split) return x_split_unsupported("hrseg_conv_wgrad");
  HRSEG_CHECK_ARG(s_in != nullptr, "hrseg_conv_wgrad: null shape");
  hrseg_conv_shape_t sh;
  resolve_wgrad_shapes(1, s_in, &sh);
  const hrseg_conv_shape_t* s = &sh;
  if (int e = check_shape(s, "hrseg_conv_wgrad")) return e;
  HRSEG_CHECK_ARG(x && dy && dw, "hrseg_conv_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int T = s->ksize * s->ksize;
  if (s->Cin <= HRSEG_SMALL_CIN_MAX) {
    const long M = (long)s->B * s->Ho * s->Wo;
    // ~512 blocks (measured, tools/misc_bench.py): every block adds into the same Cout*T*Cin
    // addresses, ~0.35 us of serialized atomics per block; fewer blocks leave CUs idle
    const int nblk_target = hrseg_g_deterministic ? 1 : g_tune_wg_blocks ? g_tune_wg_blocks : 512;
    int ppb = (int)ceil_div(ceil_div(M, nblk_target), 64) * 64;
    launch_small_cin_wgrad(x, dy, dw, s, ppb, st);
    ++g_cnt[CNT_SMALL_CIN];
    HRSEG_LAUNCH_CHECK("conv_small_cin_wgrad");
    return 0;
  }
  HRSEG_CHECK_ARG(s->Cin % 16 == 0 && s->Cout % 16 == 0, "hrseg_conv_wgrad: channels (%d,%d) must be multiples of 16",
                  s->Cin, s->Cout);
  HRSEG_CHECK_ARG(s->ldx % 4 == 0 && s->ldy % 4 == 0, "hrseg_conv_wgrad: ld must be a multiple of 4");
  WgradArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.ldx = s->ldx; a.lddy = s->ldy;
  a.B = s->B; a.Hi = s->Hi; a.Wi = s->Wi; a.Cin = s->Cin; a.Ho = s->Ho; a.Wo = s->Wo; a.Cout = s->Cout;
  a.M = s->B * s->Ho * s->Wo; a.ks = s->ksize; a.stride = s->stride; a.T = T;
  const bool big = (long)s->B * s->Ho * s->Wo >= (1L << 24);   // float-reciprocal division is exact below 2^24
  a.rcp_hw = big ? 0.f : 1.0f / (float)(s->Ho * s->Wo);
  a.rcp_w = big ? 0.f : 1.0f / (float)s->Wo;
  a.dymax = s->precision == HRSEG_CONV_FP16X2 ? s->grad_absmax : nullptr;
  if (const int ns = sp_pieces(s->precision)) {
    if (int e = dispatch_wgrad_sp(ns, a, st)) return e;
    HRSEG_LAUNCH_CHECK("wgrad_sp");
    return 0;
  }
  const int tn = (s->Cout % 48 == 0) ? 3 : (s->Cout % 64 == 0) ? 4 : (s->Cout % 32 == 0) ? 2 : 1;
  const int tk = (s->Cin % 48 == 0) ? 3 : (s->Cin % 64 == 0) ? 4 : (s->Cin % 32 == 0) ? 2 : 1;
  {
    // measured (tools/wgrad_sweep.py): 64-pixel stages, single LDS buffer; grid of ~7 blocks per
    // output tile set, between 2 and 16 blocks per CU
    const int tiles = (a.Cout / (16 * tn)) * (a.Cin / (16 * tk)) * a.T;
    int pix = 64, db = 1, target = 7 * tiles;
    if (target < 512) target = 512;
    if (target > 4096) target = 4096;
    if (g_tune_wg_pix) pix = g_tune_wg_pix;
    if (g_tune_wg_db) db = g_tune_wg_db;
    if (g_tune_wg_blocks) target = g_tune_wg_blocks;
    if (int e = launch_wgrad_f32(a, tn, tk, pix, db, target, st)) return e;
    ++g_cnt[CNT_WGRAD_F32];
  }
  HRSEG_LAUNCH_CHECK("wgrad");
  return 0;
}

extern "C" int hrseg_weight_transpose(const float* w, float* wt, int Cout, int taps, int Cin,
                                      hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(w && wt && Cout > 0 && taps > 0 && Cin > 0, "hrseg_weight_transpose: bad arguments");
  dim3 grid(ceil_div(Cin, 32), ceil_div(Cout, 32), taps);
  hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, w, wt, Cout, taps, Cin);
  HRSEG_LAUNCH_CHECK("weight_transpose");
  return 0;
}

extern "C" int hrseg_weight_transpose_all(const float* flat, float* flat_t, const int* table, int nentries,
                                          hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(flat && flat_t && table && nentries > 0, "hrseg_weight_transpose_all: bad arguments");
  hipLaunchKernelGGL(weight_transpose_all_kernel, dim3(nentries), dim3(32, 8), 0, (hipStream_t)stream, flat, flat_t,
                     table);
  HRSEG_LAUNCH_CHECK("weight_transpose_all");
  return 0;
}

// --------------------------------------------------------------------------- tuning knobs
// One entry point for the tile-plan overrides the sweep tools under tools/ use (0 = automatic plan):
//   igemm_wtm / igemm_kc / igemm_db / igemm_ksplit   fp32 implicit GEMM: pixel tiles per wave (1,2,4; +10 = 96-channel
//                                                     tiles), 16-channel chunks per stage, LDS buffers, split-K factor
//   group_wtm                                         grouped launches: 1 = 64-pixel, 2 = 128-pixel tiles
//   wgrad_pix / wgrad_db / wgrad_blocks               weight gradient: pixels per stage, LDS buffers, target grid
//   wgrad_group_mult / _min / _max                    grouped weight gradient: blocks per problem = clamp(mult*tiles, min, max)
extern "C" int hrseg_tune(const char* key, int value) {
  struct { const char* k; int* v; } tab[] = {
      {"igemm_wtm", &g_tune_wtm}, {"igemm_kc", &g_tune_kc}, {"igemm_db", &g_tune_db}, {"igemm_ksplit", &g_tune_ksplit},
      {"group_wtm", &g_group_wtm}, {"wgrad_pix", &g_tune_wg_pix}, {"wgrad_db", &g_tune_wg_db},
      {"wgrad_blocks", &g_tune_wg_blocks}, {"wgrad_group_mult", &g_wg_mult}, {"wgrad_group_min", &g_wg_min},
      {"wgrad_group_max", &g_wg_max}, {"sp_wtm", &g_sp_wtm}, {"sp_wtn", &g_sp_wtn}, {"sp_ksplit", &g_sp_ksplit}, {"sp_patch", &g_sp_patch}, {"sp_persist", &g_sp_persist}, {"sp_ws", &g_sp_ws}, {"sp_ws_waste", &g_ws_waste}, {"small_cin3", &g_small_cin3}, {"sp_ws_bf16", &g_ws_bf16}, {"sp_ws_n48", &g_ws_n48}, {"sp_img", &g_sp_img}, {"wgrad9", &g_wg9}, {"ws_epi_early", &g_ws_epi_early}, {"exp_nosplit_x", &g_exp_nosplit}, {"x_split", &g_x_split}, {"ws_epi_cost", &g_ws_epi_cost}, {"ws_epi_acc_cost", &g_ws_epi_acc_cost}, {"wgrad9_blocks", &g_wg9_blocks}, {"wgrad9_blocks1", &g_wg9_blocks_n[1]}, {"wgrad9_blocks2", &g_wg9_blocks_n[2]}, {"wgrad9_blocks3", &g_wg9_blocks_n[3]}, {"wgrad9_blocks4", &g_wg9_blocks_n[4]}, {"sp_wide", &g_sp_wide}, {"sp_ws_canvas", &g_ws_canvas}, {"wgrad_group_sp", &g_wg_group_sp}, {"wgrad_sp_t5", &g_wg_t5}, {"wgrad_sp_wide", &g_wg_wide}, {"sp_wide_min_blocks", &g_spw_min_blocks},
      {"sp_patch_min_tiles", &g_patch_min_tiles}, {"auto_min_pixels", &g_auto_min_pix}, {"sp_ws_min_tiles", &g_ws_min_tiles},
      {"deterministic", &hrseg_g_deterministic}};
  HRSEG_CHECK_ARG(key != nullptr, "hrseg_tune: null key");
  for (auto& e : tab)
    if (!strcmp(e.k, key)) {
      *e.v = value;
      if (g_patch_min_tiles <= 0) g_patch_min_tiles = 192;      // (0 = the default plan, as for every key)
      if (g_auto_min_pix <= 0) g_auto_min_pix = 8192;
      if (g_ws_min_tiles <= 0) g_ws_min_tiles = 96;
      return 0;
    }
  hrseg_set_error("hrseg_tune: unknown key '%s'", key);
  return HRSEG_ERR_INVALID_ARG;
}

