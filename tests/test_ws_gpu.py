"""The wave-specialised halo-patch kernels (csrc/conv_sp.h: igemm_patch_ws_body, sp_weight_image_kernel) through
the C ABI.

The wave-specialised body adds up the same products in the same order as the block-synchronous halo-patch body
(same K stages, slabs and per-accumulator product order; the weights are split by the same function, once instead of
per tile), so wherever both apply their results must be IDENTICAL bit for bit -- the strongest statement available,
and the one asserted here; the fp32 torch-CPU convolution pins both within the fp16x2 tolerance (2e-5 of the
reference's max magnitude).  Covered: every tiling (48, 96 and 64 output channels per tile, 8- and 16-row tiles),
ragged image edges, one and several K stages, bias, accumulate, forward and data-gradient tap geometry, the grouped
launch with its block partition, images whose tiles are mostly padding (20 x 20), the scratch ring wrapping around,
and a scratch buffer too small for the weight image (the launch then takes the other kernel)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5
# Cin, Cout, H, W, B, n48 (hrseg_tune sp_ws_n48: 0 keeps the 48-channel tilings off), exact: the block-synchronous
# halo-patch body takes the same problem when the wave-specialised one is switched off (bit-identical results);
# otherwise the im2col body does, whose reduction runs tap-major over all channels (same values to 2e-5)
WS_CASES = [
    (96, 96, 78, 78, 4, 0, True),       # 96-channel tiles, two K stages
    (192, 192, 39, 39, 8, 0, False),    # four K stages, 26 % tile padding
    (384, 384, 20, 20, 8, 0, False),    # eight K stages, tiles mostly padding
    (64, 64, 70, 61, 6, 0, True),       # 64-channel tiles / 64-channel K stages, ragged edges
    (128, 64, 62, 78, 5, 0, True),      # 64 x 64 tiling, two K stages
    (128, 64, 57, 66, 4, 0, False),     # the same with 36 % tile padding
    (48, 48, 155, 155, 2, 1, True),     # 48-channel tiles on 8-row tiles
    (48, 48, 152, 155, 4, 1, True),     # 48-channel tiles on 16-row tiles
    (96, 48, 37, 45, 13, 1, True),      # 48-channel tiles, two K stages, ragged edges
    (48, 96, 46, 92, 6, 1, True),       # too few 96-channel tiles: 48-channel tiling of a 96-channel layer
]


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


@pytest.fixture(autouse=True)
def _restore_switches():
    from hrseg_amd import _lib
    yield
    _lib.tune(sp_ws=1, sp_ws_n48=1)
    _lib.set_deterministic(False)


def _both(fn):
    """fn() with the wave-specialised body enabled and disabled"""
    from hrseg_amd import _lib
    _lib.tune(sp_ws=1)
    a = fn()
    _lib.tune(sp_ws=0)
    b = fn()
    _lib.tune(sp_ws=1)
    return a, b


@pytest.mark.parametrize("case", WS_CASES)
def test_ws_conv_bit_identical_to_block_synchronous_kernel_and_close_to_torch(case):
    from hrseg_amd import _lib, ops
    cin, cout, H, W, B, n48, exact = case
    _lib.tune(sp_ws_n48=n48)
    pr = _lib.CONV_PRECISION["fp16x2"]
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, bias, stride=1, padding=1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd, wd, dyd, bd = _nhwc(x), w.permute(0, 2, 3, 1).contiguous().cuda(), _nhwc(dy), bias.cuda()
    wt = ops.weight_transpose(wd.reshape(cout, 9, cin), cout, 9, cin)
    gmax = dyd.abs().max().reshape(1).repeat(64)

    y_ws, y_bs = _both(lambda: ops.conv_fwd(xd, wd.reshape(cout, 9, cin), bd, 3, 1, prec=pr))
    assert not exact or torch.equal(y_ws, y_bs), "forward: wave-specialised and block-synchronous results differ"
    assert _rel(y_ws, y_bs) < TOL
    assert _rel(y_ws.permute(0, 3, 1, 2), y_ref) < TOL

    dx_ws, dx_bs = _both(lambda: ops.conv_dgrad(dyd, wt, xd.shape, 3, 1, prec=pr, gmax=gmax))
    assert not exact or torch.equal(dx_ws, dx_bs), "data gradient: wave-specialised and block-synchronous results differ"
    assert _rel(dx_ws, dx_bs) < TOL
    assert _rel(dx_ws.permute(0, 3, 1, 2), xr.grad) < TOL

    # accumulate: dx += on top of an existing gradient
    base = torch.randn(xd.shape, generator=torch.Generator().manual_seed(3)).cuda()
    acc = ops.conv_dgrad(dyd, wt, xd.shape, 3, 1, out=base.clone(), accumulate=True, prec=pr, gmax=gmax)
    assert _rel((acc - base).permute(0, 3, 1, 2), xr.grad) < 2 * TOL


# Cin, Cout, H, W, B, n48: small images, where the batch is tiled as ONE canvas (images side by side, a zero column between)
CANVAS_CASES = [
    (192, 192, 39, 39, 8, 0),     # the third HRNet branch at the headline size: 1.26x -> 1.05x padded area
    (384, 384, 20, 20, 8, 0),     # the fourth: 1.92x -> 1.32x
    (64, 64, 20, 23, 3, 0),       # an image boundary inside a tile, odd batch
    (128, 64, 33, 13, 16, 0),     # pitch 14: every tile straddles an image boundary
    (64, 128, 9, 7, 11, 0),       # pitch 8: two images per tile, ragged last tile
    (96, 48, 17, 9, 13, 1),       # 48-channel tiles, two K stages
    (48, 48, 39, 39, 2, 1),       # two images
]


@pytest.mark.parametrize("case", CANVAS_CASES)
def test_ws_canvas_tiling_bit_identical_to_per_image_tiling(case):
    """canvas mode changes which tile computes a pixel, never the products or their order: forward (with bias, residual and
    ReLU in the epilogue), data gradient and accumulating data gradient must equal the per-image tiling bit for bit, and the
    fp32 torch-CPU convolution within the fp16x2 tolerance; the launch counter proves the canvas layout was taken"""
    from hrseg_amd import _lib, ops
    cin, cout, H, W, B, n48 = case
    _lib.tune(sp_ws_n48=n48, sp_ws_waste=1000, sp_ws_min_tiles=1)
    pr = _lib.CONV_PRECISION["fp16x2"]
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, bias, stride=1, padding=1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd, wd, dyd, bd, rd = _nhwc(x), w.permute(0, 2, 3, 1).contiguous().cuda().reshape(cout, 9, cin), _nhwc(dy), bias.cuda(), _nhwc(res)
    wt = ops.weight_transpose(wd, cout, 9, cin)
    gmax = dyd.abs().max().reshape(1).repeat(64)
    base = torch.randn(xd.shape, generator=torch.Generator().manual_seed(3)).cuda()

    def run():
        _lib.launch_count(None, reset=True)
        y = ops.conv_fwd(xd, wd, bd, 3, 1, prec=pr)
        yr = ops.conv_fwd(xd, wd, bd, 3, 1, prec=pr, residual=rd, relu=True)
        dx = ops.conv_dgrad(dyd, wt, xd.shape, 3, 1, prec=pr, gmax=gmax)
        acc = ops.conv_dgrad(dyd, wt, xd.shape, 3, 1, out=base.clone(), accumulate=True, prec=pr, gmax=gmax)
        return (y, yr, dx, acc), _lib.launch_count("ws"), _lib.launch_count("ws_canvas")

    try:
        _lib.tune(sp_ws_canvas=5)
        on, n_ws, n_cv = run()
        assert n_ws == 4 and n_cv == 4, (n_ws, n_cv)
        _lib.tune(sp_ws_canvas=0)
        off, n_ws, n_cv = run()
        assert n_ws == 4 and n_cv == 0, (n_ws, n_cv)
    finally:
        _lib.tune(sp_ws_canvas=5, sp_ws_waste=200, sp_ws_min_tiles=0)
    for a, b, what in zip(on, off, ("forward", "forward + residual + relu", "data gradient", "accumulating data gradient")):
        assert torch.equal(a, b), f"{what}: canvas and per-image tilings differ"
    assert _rel(on[0].permute(0, 3, 1, 2), y_ref) < TOL
    assert _rel(on[1].permute(0, 3, 1, 2), torch.relu(y_ref.detach() + res)) < TOL
    assert _rel(on[2].permute(0, 3, 1, 2), xr.grad) < TOL
    assert _rel((on[3] - base).permute(0, 3, 1, 2), xr.grad) < 2 * TOL


def _branches(B, seed=5):
    g = torch.Generator().manual_seed(seed)
    chans, sizes = [48, 96, 192, 384], [155, 78, 39, 20]
    xs = [torch.randn(B, h, h, c, generator=g).cuda() for c, h in zip(chans, sizes)]
    ws = [(torch.randn(c, 9, c, generator=g) / (9 * c) ** 0.5).cuda() for c in chans]
    return chans, xs, ws


@pytest.mark.parametrize("n48", [0, 1])
@pytest.mark.parametrize("n", [2, 3, 4])
def test_ws_group_launch_matches_single_launches(n, n48):
    """the parallel HRNet branches as the engine issues them (one grouped call, `auto` arithmetic): whatever mix of
    kernels the group dispatch picks, every branch must equal its own single fp16x2 launch"""
    from hrseg_amd import _lib, ops
    _lib.tune(sp_ws_n48=n48)
    chans, xs, ws = _branches(8)            # the batch of the headline configuration: two level passes of four images
    auto, f16 = _lib.CONV_PRECISION["auto"], _lib.CONV_PRECISION["fp16x2"]
    ys = ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n], prec=auto)
    for i in range(n):
        single = ops.conv_fwd(xs[i], ws[i], None, 3, 1, prec=f16)
        exact = ops.conv_fwd(xs[i], ws[i], None, 3, 1, prec=0)
        assert _rel(ys[i], exact) < TOL, i
        assert torch.equal(ys[i], single), f"branch {i}: grouped and single launches differ"
    # data gradient of the same group
    dys = [torch.randn_like(x) for x in xs[:n]]
    wts = [ops.weight_transpose(w, c, 9, c) for w, c in zip(ws[:n], chans[:n])]
    gms = [d.abs().max().reshape(1).repeat(64) for d in dys]
    dxs = ops.conv_dgrad_group(dys, wts, [x.shape for x in xs[:n]], 3, 1, [None] * n, [False] * n, prec=auto, gmaxs=gms)
    for i in range(n):
        exact = ops.conv_dgrad(dys[i], wts[i], xs[i].shape, 3, 1, prec=0)
        assert _rel(dxs[i], exact) < TOL, i


def test_ws_is_bit_reproducible_and_independent_of_the_block_partition():
    """no atomics, fixed summation order: repeated launches agree bit for bit, and so do a branch's results inside
    groups of different sizes (different numbers of persistent blocks per problem)"""
    from hrseg_amd import _lib, ops
    chans, xs, ws = _branches(8, seed=9)
    auto = _lib.CONV_PRECISION["auto"]
    first = ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans, prec=auto)
    again = ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans, prec=auto)
    two = ops.conv_fwd_group(xs[2:], ws[2:], [None] * 2, 3, 1, chans[2:], prec=auto)
    for i in range(4):
        assert torch.equal(first[i], again[i])
    for i in range(2):
        assert torch.equal(first[i + 2], two[i])


def test_scratch_ring_wraps_and_small_scratch_falls_back():
    """weight images go through a ring in the caller's scratch buffer: launches with changing weights must each
    see their own image after the ring has wrapped; a buffer too small for the image makes the launch take the
    block-synchronous kernel (same bits)"""
    from hrseg_amd import _lib, ops
    pr = _lib.CONV_PRECISION["fp16x2"]
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 78, 78, 96, generator=g).cuda()
    wlist = [(torch.randn(96, 9, 96, generator=g) / 30).cuda() for _ in range(7)]
    _lib.tune(sp_ws=0)
    want = [ops.conv_fwd(x, w, None, 3, 1, prec=pr) for w in wlist]
    _lib.tune(sp_ws=1)
    image_bytes = 96 * 9 * 96 * 4
    try:
        # eight regions of 0.75 MiB: two images of 324 KiB fit, the third wraps
        small = torch.empty(6 << 20, dtype=torch.uint8, device="cuda")
        _lib.call_raw("hrseg_set_scratch", small.data_ptr(), small.numel())
        assert (small.numel() // 8) // image_bytes == 2
        for _ in range(3):
            for w, y in zip(wlist, want):
                assert torch.equal(ops.conv_fwd(x, w, None, 3, 1, prec=pr), y)
        # 1 MiB: regions of 128 KiB cannot hold the image
        tiny = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
        _lib.call_raw("hrseg_set_scratch", tiny.data_ptr(), tiny.numel())
        assert torch.equal(ops.conv_fwd(x, wlist[0], None, 3, 1, prec=pr), want[0])
        _lib.call_raw("hrseg_set_scratch", None, 0)
        assert torch.equal(ops.conv_fwd(x, wlist[1], None, 3, 1, prec=pr), want[1])
    finally:
        torch.cuda.synchronize()
        _lib._scratch = None            # ops re-attaches the default buffer at the next convolution
        _lib.ensure_scratch(x.device)


@pytest.mark.parametrize("mb", [64, 48])
def test_group_images_never_wrap_inside_a_group(mb):
    """the four-branch group (86 KB + 344 KB + 1.3 MB + 5.3 MB of weight images) with a small scratch buffer: with
    8 MiB regions (64 MiB) the group fits once and every later group wraps BEFORE its first image; with 6 MiB regions
    (48 MiB) the group does not fit and the launch takes its other kernels.  Either way forward and data gradient must
    equal the exact-fp32 kernels' results over several rounds with changing weights (a wrap inside a group would
    overwrite an earlier image of the same launch: the round-2 defect)."""
    from hrseg_amd import _lib, ops
    auto = _lib.CONV_PRECISION["auto"]
    chans, xs, _ = _branches(8, seed=13)
    g = torch.Generator().manual_seed(77)
    rounds = [[(torch.randn(c, 9, c, generator=g) / (9 * c) ** 0.5).cuda() for c in chans] for _ in range(4)]
    dys = [torch.randn_like(x) for x in xs]
    gms = [d.abs().max().reshape(1).repeat(64) for d in dys]
    try:
        small = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
        _lib.call_raw("hrseg_set_scratch", small.data_ptr(), small.numel())
        _lib.launch_count(None, reset=True)
        for ws in rounds:
            ys = ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans, prec=auto)
            wts = [ops.weight_transpose(w, c, 9, c) for w, c in zip(ws, chans)]
            dxs = ops.conv_dgrad_group(dys, wts, [x.shape for x in xs], 3, 1, [None] * 4, [False] * 4, prec=auto, gmaxs=gms)
            for i in range(4):
                assert _rel(ys[i], ops.conv_fwd(xs[i], ws[i], None, 3, 1, prec=0)) < TOL, i
                assert _rel(dxs[i], ops.conv_dgrad(dys[i], wts[i], xs[i].shape, 3, 1, prec=0)) < TOL, i
        n_group = _lib.launch_count("ws_group")
        assert (n_group == 8) if mb == 64 else (n_group == 0), n_group
    finally:
        torch.cuda.synchronize()
        _lib._scratch = None
        _lib.ensure_scratch(xs[0].device)


def test_fp16x2_out_of_range_activations_are_loud_and_guarded():
    """fp16x2 takes its activation operand unscaled: far beyond the fp16 range the result must be Inf / NaN (never a
    silently saturated finite value), NaN inputs must propagate, and in deterministic mode ops.py range-checks the
    operand (hrseg_absmax) and runs the exact-fp32 kernels instead"""
    from hrseg_amd import _lib, ops
    pr = _lib.CONV_PRECISION["fp16x2"]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 40, 48, 96, generator=g).cuda()
    w = (torch.randn(96, 9, 96, generator=g) / 30).cuda()
    x_big = x.clone()
    x_big[0, 7, 9, 5] = 3.0e5                              # > 2 * 65504: not representable by the two fp16 pieces
    want = ops.conv_fwd(x_big, w, None, 3, 1, prec=0)
    got = ops.conv_fwd(x_big, w, None, 3, 1, prec=pr)
    assert not bool(torch.isfinite(got).all()), "an out-of-range activation must not give a finite (saturated) result"
    x_nan = x.clone()
    x_nan[1, 3, 3, 0] = float("nan")
    assert bool(torch.isnan(ops.conv_fwd(x_nan, w, None, 3, 1, prec=pr)[1, 3, 3]).all()), "NaN must propagate"
    assert float(ops.absmax(x_nan)) == float("inf") and abs(float(ops.absmax(x)) - float(x.abs().max())) == 0.0
    _lib.set_deterministic(True)
    try:
        before, _ = ops.range_fallbacks, _lib.launch_count(None, reset=True)
        got = ops.conv_fwd(x_big, w, None, 3, 1, prec=pr)
        assert ops.range_fallbacks == before + 1 and _lib.launch_count("f32") == 1 and _lib.launch_count("ws") == 0
        assert torch.equal(got, want)
        ok = ops.conv_fwd(x, w, None, 3, 1, prec=pr)            # in range: stays on the fp16x2 kernels
        assert ops.range_fallbacks == before + 1 and _rel(ok, ops.conv_fwd(x, w, None, 3, 1, prec=0)) < TOL
    finally:
        _lib.set_deterministic(False)


@pytest.mark.parametrize("cfg", ["branches1", "branches3", "branches4", "tiles64"])
def test_nine_tap_weight_gradient_matches_torch_and_is_bit_reproducible(cfg):
    """the nine-tap weight gradient (per-block partial sums in a workspace + ordered reduce, no atomics) on the grouped branch
    launches (48-channel tiling) and on 64-channel tilings (UNet, the HRNet stem stage): ragged image edges, accumulation into
    an existing gradient, two runs give the same bits, values against torch"""
    from hrseg_amd import _lib, ops
    pr = _lib.CONV_PRECISION["fp16x2"]
    g = torch.Generator().manual_seed(40 + len(cfg))
    if cfg == "tiles64":
        shapes = [(64, 128, 37, 45, 3), (128, 64, 18, 23, 3), (256, 256, 9, 12, 3)]
    else:
        n = int(cfg[-1])
        shapes = [(c, c, h, w, 5) for c, (h, w) in zip([48, 96, 192, 384][:n], [(61, 83), (31, 42), (16, 21), (8, 11)][:n])]
    xs = [torch.randn(b, h, w, ci, generator=g).cuda() for ci, co, h, w, b in shapes]
    dys = [(torch.randn(b, h, w, co, generator=g) * 1e-3).cuda() for ci, co, h, w, b in shapes]
    gms = [d.abs().max().reshape(1).repeat(64) for d in dys]
    base = [torch.randn(co, 9, ci, generator=g).cuda() for ci, co, h, w, b in shapes]
    outs = []
    for _ in range(2):
        dws = [b.clone() for b in base]
        _lib.launch_count(None, reset=True)
        ops.conv_wgrad_group(xs, dys, dws, 3, 1, prec=pr, gmaxs=gms)
        assert _lib.launch_count("wgrad9") == 1
        outs.append(dws)
    for a, a2, x, dy, b0, (ci, co, h, w, b) in zip(outs[0], outs[1], xs, dys, base, shapes):
        assert torch.equal(a, a2), "two runs of the nine-tap weight gradient differ"
        wr = torch.zeros(co, ci, 3, 3, requires_grad=True)
        F.conv2d(x.permute(0, 3, 1, 2).cpu(), wr, padding=1).backward(dy.permute(0, 3, 1, 2).cpu())
        assert _rel((a - b0).view(co, 3, 3, ci).permute(0, 3, 1, 2).cpu(), wr.grad) < 4e-5


@pytest.mark.parametrize("cfg", ["branches4", "branches2_bias", "single64", "single_canvas"])
def test_epilogue_statistics_are_the_sums_of_the_output(cfg):
    """BatchNorm partial sums from the convolution epilogue (hrseg_conv_shape_t.stat_partial): per problem the rows the call
    reports add up to (sum y, sum y^2) per output channel of the tensor it wrote -- grouped launches with their block
    partition, ragged tiles, canvas-tiled small images, bias; and BatchNorm on those rows (phases 6) equals BatchNorm with
    its own statistics launch"""
    from hrseg_amd import _lib, ops
    pr = _lib.CONV_PRECISION["auto"]          # (grouped launches reach the wave-specialised kernels through the default policy)
    g = torch.Generator().manual_seed(7)
    if cfg == "branches4":
        shapes = [(48, 48, 61, 83, 4), (96, 96, 31, 42, 4), (192, 192, 16, 21, 4), (384, 384, 8, 11, 4)]
    elif cfg == "branches2_bias":
        shapes = [(48, 48, 77, 77, 2), (96, 96, 39, 39, 2)]
    elif cfg == "single64":
        shapes = [(64, 128, 45, 52, 3)]
    else:
        shapes = [(192, 192, 20, 20, 8)]
    bias = cfg == "branches2_bias"
    xs = [torch.randn(b, h, w, ci, generator=g).cuda() for ci, co, h, w, b in shapes]
    ws = [(torch.randn(co, 9, ci, generator=g) * 0.05).cuda() for ci, co, h, w, b in shapes]
    bs = [(torch.randn(co, generator=g)).cuda() if bias else None for ci, co, h, w, b in shapes]
    couts = [co for ci, co, h, w, b in shapes]
    try:
        _lib.tune(sp_ws_min_tiles=1)
        _lib.launch_count(None, reset=True)
        if len(shapes) == 1:
            y, st = ops.conv_fwd(xs[0], ws[0], bs[0], 3, 1, prec=pr, stats=True)
            ys, stats = [y], [st]
        else:
            ys, stats = ops.conv_fwd_group(xs, ws, bs, 3, 1, couts, prec=pr, stats=True)
        assert _lib.launch_count("ws") + _lib.launch_count("ws_group") == 1
        if cfg == "single_canvas":
            assert _lib.launch_count("ws_canvas") == 1
    finally:
        _lib.tune(sp_ws_min_tiles=0)
    for y, st, co in zip(ys, stats, couts):
        assert st is not None, "the wave-specialised launch reported no statistics rows"
        part, rows = st
        assert 0 < rows <= 256
        tot = part[:rows * 2 * co].view(rows, 2, co).sum(0).cpu()
        y64 = y.double().reshape(-1, co).cpu()
        s1, s2 = y64.sum(0), (y64 * y64).sum(0)
        assert float((tot[0] - s1).abs().max()) < 1e-5 * float(s2.sqrt().max()) * 30, cfg       # |sum| errors scale with sqrt(n) |y|
        assert float(((tot[1] - s2) / s2).abs().max()) < 2e-6, cfg
    # BatchNorm on those rows == BatchNorm with its own statistics phase
    def bn_items(parts):
        return [dict(y=y, gamma=torch.ones(co, device="cuda"), beta=torch.zeros(co, device="cuda"),
                     rm=torch.zeros(co, device="cuda"), rv=torch.ones(co, device="cuda"),
                     nbt=torch.zeros((), dtype=torch.int64, device="cuda"), momentum=0.1, eps=1e-5, residual=None, relu=True,
                     partial=p) for y, co, p in zip(ys, couts, parts)]
    a_items, b_items = bn_items(stats), bn_items([None] * len(ys))
    za = ops.bn_fwd_group(a_items, True, phases=6)
    zb = ops.bn_fwd_group(b_items, True)
    for (z1, c1), (z2, c2), ia, ib in zip(za, zb, a_items, b_items):
        assert float((z1 - z2).abs().max()) < 2e-5 * max(1.0, float(z2.abs().max()))
        assert float((ia["rv"] - ib["rv"]).abs().max()) < 1e-6 and float((ia["rm"] - ib["rm"]).abs().max()) < 1e-6


BF16_TOL = 6e-3        # operands rounded to bf16 (2^-9 each), fp32 accumulation: 2.2-2.8e-3 measured on these shapes


@pytest.mark.parametrize("case", [WS_CASES[0], WS_CASES[1], WS_CASES[3], WS_CASES[6], WS_CASES[8]])
def test_bf16_instance_of_the_wave_specialised_body(case):
    """BASELINE configs[4] arithmetic (conv_dtype 'bf16': one bf16 piece per operand, one MFMA product) on the
    wave-specialised kernels: forward, data gradient and accumulating data gradient against the block-synchronous /
    im2col bf16 kernels (hrseg_tune sp_ws_bf16=0; bit-identical where the halo-patch body takes the problem) and torch"""
    from hrseg_amd import _lib, ops
    cin, cout, H, W, B, n48, exact = case
    _lib.tune(sp_ws_n48=n48)
    pr = _lib.CONV_PRECISION["bf16"]
    g = torch.Generator().manual_seed(sum(case[:5]) + 1)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, bias, stride=1, padding=1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd, wd, dyd, bd = _nhwc(x), w.permute(0, 2, 3, 1).contiguous().cuda().reshape(cout, 9, cin), _nhwc(dy), bias.cuda()
    wt = ops.weight_transpose(wd, cout, 9, cin)
    base = torch.randn(xd.shape, generator=torch.Generator().manual_seed(3)).cuda()

    def run():
        _lib.launch_count(None, reset=True)
        y = ops.conv_fwd(xd, wd, bd, 3, 1, prec=pr)
        dx = ops.conv_dgrad(dyd, wt, xd.shape, 3, 1, prec=pr)
        acc = ops.conv_dgrad(dyd, wt, xd.shape, 3, 1, out=base.clone(), accumulate=True, prec=pr)
        return (y, dx, acc), _lib.launch_count("ws")

    try:
        _lib.tune(sp_ws_bf16=1)
        on, n_on = run()
        _lib.tune(sp_ws_bf16=0)
        off, n_off = run()
    finally:
        _lib.tune(sp_ws_bf16=1)
    assert n_on == 3 and n_off == 0, (n_on, n_off)
    for a, b, what in zip(on, off, ("forward", "data gradient", "accumulating data gradient")):
        assert not exact or torch.equal(a, b), f"{what}: wave-specialised and block-synchronous bf16 results differ"
        assert _rel(a, b) < 1e-5 if exact else _rel(a, b) < BF16_TOL, what
    assert _rel(on[0].permute(0, 3, 1, 2), y_ref) < BF16_TOL
    assert _rel(on[1].permute(0, 3, 1, 2), xr.grad) < BF16_TOL
    assert _rel((on[2] - base).permute(0, 3, 1, 2), xr.grad) < 2 * BF16_TOL


def test_bf16_group_launch_on_the_wave_specialised_body():
    """the four parallel branches under conv_dtype 'bf16' as one grouped call: one wave-specialised launch (bf16 instance),
    every branch equal to its own single launch and to the exact-fp32 kernels within the bf16 tolerance"""
    from hrseg_amd import _lib, ops
    chans, xs, ws = _branches(8, seed=17)
    pr = _lib.CONV_PRECISION["bf16"]
    _lib.launch_count(None, reset=True)
    ys = ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans, prec=pr)
    assert _lib.launch_count("ws_group") == 1
    for i in range(4):
        single = ops.conv_fwd(xs[i], ws[i], None, 3, 1, prec=pr)
        assert torch.equal(ys[i], single), f"branch {i}: grouped and single bf16 launches differ"
        assert _rel(ys[i], ops.conv_fwd(xs[i], ws[i], None, 3, 1, prec=0)) < BF16_TOL
    dys = [torch.randn_like(x) for x in xs]
    wts = [ops.weight_transpose(w, c, 9, c) for w, c in zip(ws, chans)]
    _lib.launch_count(None, reset=True)
    dxs = ops.conv_dgrad_group(dys, wts, [x.shape for x in xs], 3, 1, [None] * 4, [False] * 4, prec=pr)
    assert _lib.launch_count("ws_group") == 1
    for i in range(4):
        assert _rel(dxs[i], ops.conv_dgrad(dys[i], wts[i], xs[i].shape, 3, 1, prec=0)) < BF16_TOL


@pytest.mark.gpu
def test_ws_results_do_not_depend_on_timing():
    """The producer waves' loads are inline assembly with counted waits the compiler knows nothing about (a register copied or
    reused while its load is in flight shows up as run-to-run differences, not as a wrong constant): every case 25 times, each
    result bit-identical to the first.  (tools/ws_stress.py is the long form.)"""
    from hrseg_amd import _lib, ops
    pr = _lib.CONV_PRECISION["auto"]
    g = torch.Generator(device="cuda").manual_seed(3)
    dev = torch.device("cuda:0")
    for cin, cout, H, W, B in [(48, 48, 155, 155, 8), (144, 96, 78, 61, 5), (192, 192, 39, 39, 8), (64, 64, 155, 155, 4), (96, 48, 37, 45, 13)]:
        x = torch.randn(B, H, W, cin, device=dev, generator=g)
        w = torch.randn(cout, 9, cin, device=dev, generator=g) * 0.05
        wt = ops.weight_transpose(w, cout, 9, cin)
        dy = torch.randn(B, H, W, cout, device=dev, generator=g) * 1e-3
        gm = dy.abs().max().reshape(1).repeat(64)
        y0 = ops.conv_fwd(x, w, None, 3, 1, prec=pr).clone()
        d0 = ops.conv_dgrad(dy, wt, x.shape, 3, 1, prec=pr, gmax=gm).clone()
        assert torch.isfinite(y0).all() and torch.isfinite(d0).all()
        for _ in range(25):
            assert torch.equal(ops.conv_fwd(x, w, None, 3, 1, prec=pr), y0), (cin, cout, H, W, B)
            assert torch.equal(ops.conv_dgrad(dy, wt, x.shape, 3, 1, prec=pr, gmax=gm), d0), (cin, cout, H, W, B)
    sizes, chans = [155, 78, 39, 20], [48, 96, 192, 384]
    xs = [torch.randn(8, h, h, c, device=dev, generator=g) for c, h in zip(chans, sizes)]
    ws = [torch.randn(c, 9, c, device=dev, generator=g) * 0.05 for c in chans]
    y0 = [y.clone() for y in ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans, prec=pr)]
    for _ in range(25):
        ys = ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans, prec=pr)
        assert all(torch.equal(a, b) for a, b in zip(ys, y0))
