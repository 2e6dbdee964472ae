"""Every HIP entry point against a plain fp32 torch-CPU reference of the same op (through the
C ABI, on the GPU).  Tolerances are relative to the reference's max magnitude."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from hrseg_amd import ops as o
    assert torch.cuda.is_available()
    return o


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def nhwc(x):  # NCHW cpu -> NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(x):  # NHWC cuda -> NCHW cpu
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def store(w):  # [O,I,kh,kw] -> [O][kh*kw][I] storage
    return w.permute(0, 2, 3, 1).contiguous().cuda()


CONV_CASES = [
    # Cin, Cout, k, s, H, W, B
    (48, 48, 3, 1, 37, 41, 2), (64, 128, 3, 1, 20, 20, 2), (96, 96, 3, 1, 19, 23, 3), (256, 48, 3, 1, 16, 16, 1),
    (48, 96, 3, 2, 31, 31, 2), (96, 192, 3, 2, 20, 18, 2), (64, 64, 3, 2, 33, 30, 2), (64, 256, 1, 1, 17, 17, 2),
    (144, 144, 1, 1, 15, 15, 2), (96, 48, 1, 1, 9, 9, 2), (384, 384, 3, 1, 10, 10, 2), (128, 64, 3, 1, 24, 24, 1),
    (192, 192, 3, 1, 39, 39, 4), (16, 32, 3, 1, 8, 8, 1), (3, 64, 3, 2, 30, 30, 2), (3, 64, 3, 1, 21, 17, 2),
    (48, 48, 3, 1, 155, 155, 4), (48, 96, 3, 1, 7, 5, 3), (96, 48, 3, 1, 1, 9, 2), (48, 48, 3, 1, 64, 1, 2),
    (144, 48, 3, 1, 13, 64, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(ops, case):
    cin, cout, k, s, H, W, B = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, bias, stride=s, padding=(k - 1) // 2)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)

    xd, wd = nhwc(x), store(w)
    y = ops.conv_fwd(xd, wd, bias.cuda(), k, s)
    assert rel(nchw(y), y_ref) < 2e-5
    dyd = nhwc(dy)
    dw = torch.zeros_like(wd)
    ops.conv_wgrad(xd, dyd, dw, k, s)
    dw_nchw = dw.view(cout, k, k, cin).permute(0, 3, 1, 2).cpu()
    assert rel(dw_nchw, wr.grad) < 5e-5
    # second call accumulates
    ops.conv_wgrad(xd, dyd, dw, k, s)
    assert rel(dw.view(cout, k, k, cin).permute(0, 3, 1, 2).cpu(), 2 * wr.grad) < 5e-5
    if cin % 16 == 0:
        wt = ops.weight_transpose(wd, cout, k * k, cin)
        dx = ops.conv_dgrad(dyd, wt, xd.shape, k, s)
        assert rel(nchw(dx), xr.grad) < 2e-5
        ops.conv_dgrad(dyd, wt, xd.shape, k, s, out=dx, accumulate=True)
        assert rel(nchw(dx), 2 * xr.grad) < 2e-5


def test_conv_beyond_2p24_pixels(ops):
    """tensors of 2^24 pixels or more take the integer-division index path (the float-reciprocal division is
    exact only below 2^24): forward, data gradient and weight gradient of a 3x3 conv on a 4100x4100 image"""
    H = W = 4100                                  # 16.81 M pixels > 2^24
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 16, H, W, generator=g)
    w = (torch.randn(16, 16, 3, 3, generator=g) / 12).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, padding=1)
    dy = torch.randn(1, 16, H, W, generator=g)
    y_ref.backward(dy)
    xd, wd = nhwc(x), store(w.detach())
    y = ops.conv_fwd(xd, wd, None, 3, 1)
    assert rel(nchw(y), y_ref) < 2e-5
    dyd = nhwc(dy)
    dw = torch.zeros_like(wd)
    ops.conv_wgrad(xd, dyd, dw, 3, 1)
    assert rel(dw.view(16, 3, 3, 16).permute(0, 3, 1, 2).cpu(), w.grad) < 2e-4      # 16.8 M-term fp32 sums
    dx = ops.conv_dgrad(dyd, ops.weight_transpose(wd, 16, 9, 16), xd.shape, 3, 1)
    assert rel(nchw(dx), xr.grad) < 2e-5


def test_conv_into_channel_slice(ops):
    """output written into a channel slice of a wider NHWC buffer (concat without copy)"""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 12, 12, generator=g)
    w = torch.randn(48, 32, 3, 3, generator=g) / 17
    buf = torch.zeros(2, 12, 12, 112, device="cuda")
    ops.conv_fwd(nhwc(x), store(w), None, 3, 1, out=buf[..., 64:112])
    assert rel(nchw(buf[..., 64:112]), F.conv2d(x, w, padding=1)) < 2e-5
    assert float(buf[..., :64].abs().max()) == 0.0


@pytest.mark.parametrize("C,H,W,B,relu,res", [(48, 37, 41, 2, True, True), (64, 20, 20, 3, True, False),
                                              (720, 9, 9, 2, False, False), (96, 31, 17, 2, False, True),
                                              (1024, 5, 5, 2, True, False), (192, 13, 13, 1, True, True)])
def test_batchnorm_train_fwd_bwd(ops, C, H, W, B, relu, res):
    g = torch.Generator().manual_seed(C + H)
    y = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    r = torch.randn(B, C, H, W, generator=g) if res else None
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z_ref = F.batch_norm(yr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if res:
        z_ref = z_ref + rr
    if relu:
        z_ref = F.relu(z_ref)
    dz = torch.randn(z_ref.shape, generator=g)
    z_ref.backward(dz)

    yd = nhwc(y)
    rmd, rvd, nbt = rm.cuda(), rv.cuda(), torch.zeros((), dtype=torch.int64, device="cuda")
    coef = ops.bn_train_coef(yd, gamma.cuda(), beta.cuda(), rmd, rvd, nbt, 0.1, 1e-5)
    rd = nhwc(r) if res else None
    z = ops.bn_apply(yd, coef, rd, relu)
    assert rel(nchw(z), z_ref) < 1e-5
    assert rel(rmd, rm_ref) < 1e-5 and rel(rvd, rv_ref) < 1e-5 and int(nbt) == 1
    dgam, dbet = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dres = torch.empty_like(yd) if res else None
    dy = ops.bn_bwd(nhwc(dz), z, relu, yd, coef, dgam, dbet, dres)
    scale = float(yr.grad.abs().max())
    assert float((nchw(dy) - yr.grad).abs().max()) < 2e-5 * max(scale, 1.0)
    assert rel(dgam, gr.grad) < 2e-5 and rel(dbet, br.grad) < 2e-5
    if res:
        assert rel(nchw(dres), rr.grad) < 1e-6


def test_batchnorm_eval_coef(ops):
    C = 48
    g = torch.Generator().manual_seed(0)
    y = torch.randn(2, C, 7, 9, generator=g)
    gamma, beta, rm, rv = torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.5
    coef = ops.bn_eval_coef(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5)
    z = ops.bn_apply(nhwc(y), coef, None, False)
    assert rel(nchw(z), F.batch_norm(y, rm, rv, gamma, beta, False, 0.1, 1e-5)) < 1e-5


def test_batchnorm_group_fwd_bwd(ops):
    """grouped BN (statistics, finalize, apply; reduce, totals, apply): four problems of different size per
    launch, run twice, with and without residual; without residual the backward gets no z and recomputes
    the ReLU mask from y"""
    cfgs = [(48, 37, 41, 2, True), (96, 19, 20, 2, False), (384, 5, 6, 2, False), (192, 9, 9, 3, True)]
    g = torch.Generator().manual_seed(7)
    probs = []
    for C, H, W, B, res in cfgs:
        y = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
        r = torch.randn(B, C, H, W, generator=g) if res else None
        dz = torch.randn(B, C, H, W, generator=g)
        probs.append((y, gamma, beta, r, dz))
    for rep in range(2):
        refs, items = [], []
        for y, gamma, beta, r, dz in probs:
            C = y.shape[1]
            yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            rr = r.clone().requires_grad_(True) if r is not None else None
            rm_ref, rv_ref = torch.zeros(C), torch.ones(C)
            z_ref = F.batch_norm(yr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
            if rr is not None:
                z_ref = z_ref + rr
            z_ref = F.relu(z_ref)
            z_ref.backward(dz)
            refs.append((z_ref.detach(), yr.grad, gr.grad, br.grad, rr.grad if rr is not None else None, rm_ref, rv_ref))
            items.append(dict(y=nhwc(y), gamma=gamma.cuda(), beta=beta.cuda(), rm=torch.zeros(C, device="cuda"),
                              rv=torch.ones(C, device="cuda"), nbt=torch.zeros((), dtype=torch.int64, device="cuda"),
                              momentum=0.1, eps=1e-5, residual=nhwc(r) if r is not None else None, relu=True))
        zc = ops.bn_fwd_group(items, True)
        bw = []
        for (y, gamma, beta, r, dz), it, (z, coef), ref in zip(probs, items, zc, refs):
            assert rel(nchw(z), ref[0]) < 1e-5
            assert rel(it["rm"], ref[5]) < 1e-5 and rel(it["rv"], ref[6]) < 1e-5 and int(it["nbt"]) == 1
            C = y.shape[1]
            bw.append(dict(dz=nhwc(dz), z=z if r is not None else None, relu=True, y=it["y"], coef=coef,
                           dgamma=torch.zeros(C, device="cuda"), dbeta=torch.zeros(C, device="cuda"),
                           dres=torch.empty_like(it["y"]) if r is not None else None, dres_accumulate=False))
        dys = ops.bn_bwd_group(bw, False)
        for b, dy, ref in zip(bw, dys, refs):
            scale = max(float(ref[1].abs().max()), 1.0)
            assert float((nchw(dy) - ref[1]).abs().max()) < 2e-5 * scale
            assert rel(b["dgamma"], ref[2]) < 2e-5 and rel(b["dbeta"], ref[3]) < 2e-5
            if ref[4] is not None:
                assert rel(nchw(b["dres"]), ref[4]) < 1e-6


@pytest.mark.parametrize("H,W", [(20, 20), (31, 17), (5, 8)])
def test_maxpool(ops, H, W):
    g = torch.Generator().manual_seed(H)
    x = F.relu(torch.randn(2, 64, H, W, generator=g))      # many exact ties at 0, as after ReLU
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 2)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd = nhwc(x)
    y = ops.maxpool2_fwd(xd)
    assert torch.equal(nchw(y), y_ref.detach())
    dx = ops.maxpool2_bwd(xd, nhwc(dy))
    assert torch.equal(nchw(dx), xr.grad)


@pytest.mark.parametrize("Hi,Wi,Ho,Wo,C", [(10, 10, 20, 20, 64), (20, 20, 155, 155, 48), (39, 39, 155, 155, 16),
                                           (7, 9, 14, 18, 128), (1, 1, 4, 4, 16), (8, 8, 8, 8, 32),
                                           (20, 20, 155, 155, 384), (5, 7, 78, 78, 96), (3, 3, 40, 40, 1024)])
def test_bilinear_align_corners(ops, Hi, Wi, Ho, Wo, C):
    g = torch.Generator().manual_seed(Hi + Ho)
    x = torch.randn(2, C, Hi, Wi, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=True)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd = nhwc(x)
    out = torch.empty(2, Ho, Wo, C, device="cuda")
    ops.bilinear_fwd(xd, out, Ho, Wo)
    assert rel(nchw(out), y_ref) < 1e-5
    dx = ops.bilinear_bwd(nhwc(dy), xd.shape, Ho, Wo)
    assert rel(nchw(dx), xr.grad) < 1e-5


def test_bilinear_pad_slice_accumulate(ops):
    """UNet `up`: x2 upsample, zero pad to the skip size, written into the concat slice."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 7, 7, generator=g)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    ref = F.pad(up, (0, 1, 0, 1))                         # 14 -> 15, diff//2 = 0 left/top
    buf = torch.full((2, 15, 15, 64), 7.0, device="cuda")
    ops.bilinear_fwd(nhwc(x), buf[..., 32:], 14, 14, 0, 0)
    assert rel(nchw(buf[..., 32:]), ref) < 1e-5
    assert float((buf[..., :32] - 7.0).abs().max()) == 0.0
    dbuf = torch.randn(2, 15, 15, 64, device="cuda")
    dx = ops.bilinear_bwd(dbuf[..., 32:], (2, 7, 7, 32), 14, 14, 0, 0)
    xr = x.clone().requires_grad_(True)
    F.pad(F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True), (0, 1, 0, 1)).backward(
        nchw(dbuf[..., 32:]))
    assert rel(nchw(dx), xr.grad) < 1e-5
    # accumulate + relu (HRNet fuse sum)
    base = torch.randn(2, 14, 14, 32, device="cuda")
    out = base.clone()
    ops.bilinear_fwd(nhwc(x), out, 14, 14, accumulate=True, relu=True)
    assert rel(nchw(out), F.relu(nchw(base) + up)) < 1e-5


def test_glue_kernels(ops):
    a, b = torch.randn(2, 9, 9, 48, device="cuda"), torch.randn(2, 9, 9, 48, device="cuda")
    assert torch.allclose(ops.add(a, b, relu=True), F.relu(a + b))
    dst = torch.zeros(2, 9, 9, 96, device="cuda")
    ops.copy(a, dst[..., 48:])
    ops.copy(b, dst[..., 48:], accumulate=True)
    assert torch.allclose(dst[..., 48:], a + b) and float(dst[..., :48].abs().max()) == 0
    assert torch.equal(ops.relu_bwd(a, b), torch.where(b > 0, a, torch.zeros_like(a)))
    x = torch.randn(2, 3, 11, 13)
    assert torch.equal(ops.nchw_to_nhwc(x.cuda()).cpu(), x.permute(0, 2, 3, 1).contiguous())
    z = torch.randn(2, 11, 13, 4, device="cuda")
    assert torch.equal(ops.nhwc_to_nchw(z), z.permute(0, 3, 1, 2).contiguous())


@pytest.mark.parametrize("F_,Cout,film", [(64, 4, True), (64, 7, False), (720, 4, True), (720, 3, True), (64, 2, True)])
def test_head_fwd_bwd(ops, F_, Cout, film):
    g = torch.Generator().manual_seed(F_ + Cout)
    B, H, W = 2, 13, 11
    f = torch.randn(B, F_, H, W, generator=g)
    w = torch.randn(Cout, F_, generator=g) / F_ ** 0.5
    bias = torch.randn(Cout, generator=g)
    gb = torch.randn(B, 2 * F_, generator=g) if film else None
    fr, wr, br = f.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    gbr = gb.clone().requires_grad_(True) if film else None
    fm = fr * gbr[:, :F_, None, None] + gbr[:, F_:, None, None] if film else fr
    z_ref = F.conv2d(fm, wr[:, :, None, None], br)
    dz = torch.randn(z_ref.shape, generator=g)
    z_ref.backward(dz)
    fd = nhwc(f)
    gbd = gb.cuda() if film else None
    z = ops.head_fwd(fd, gbd, w.cuda(), bias.cuda())
    assert rel(nchw(z), z_ref) < 1e-5
    dw, db = torch.zeros(Cout, F_, device="cuda"), torch.zeros(Cout, device="cuda")
    dgb = torch.zeros(B, 2 * F_, device="cuda") if film else None
    df = ops.head_bwd(fd, gbd, w.cuda(), nhwc(dz), dw, db, dgb)
    assert rel(nchw(df), fr.grad) < 1e-5
    assert rel(dw, wr.grad) < 2e-5 and rel(db, br.grad) < 2e-5
    if film:
        assert rel(dgb, gbr.grad) < 2e-5


def test_logits_up(ops):
    g = torch.Generator().manual_seed(9)
    z = torch.randn(2, 4, 16, 16, generator=g)
    zr = z.clone().requires_grad_(True)
    ref = F.interpolate(zr, size=(62, 62), mode="bilinear", align_corners=True)
    d = torch.randn(ref.shape, generator=g)
    ref.backward(d)
    out = ops.logits_up_fwd(nhwc(z), 62, 62)
    assert rel(out, ref) < 1e-5
    din = ops.logits_up_bwd(d.cuda(), 16, 16)
    assert rel(nchw(din), zr.grad) < 1e-5


def test_gap_film_linear(ops):
    g = torch.Generator().manual_seed(1)
    p = torch.rand(3, 4, 37, 29, generator=g)
    cond = ops.gap_nchw(p.cuda())
    assert rel(cond, p.mean((2, 3))) < 1e-6
    wl, bl = torch.randn(128, 4, generator=g), torch.randn(128, generator=g)
    cr, wr, br = p.mean((2, 3)).requires_grad_(True), wl.clone().requires_grad_(True), bl.clone().requires_grad_(True)
    gb_ref = F.linear(cr, wr, br)
    dgb = torch.randn(gb_ref.shape, generator=g)
    gb_ref.backward(dgb)
    gb = ops.film_linear_fwd(cond, wl.cuda(), bl.cuda())
    assert rel(gb, gb_ref) < 1e-5
    dwl, dbl = torch.zeros(128, 4, device="cuda"), torch.zeros(128, device="cuda")
    dcond = ops.film_linear_bwd(cond, wl.cuda(), dgb.cuda(), dwl, dbl, 0.5)
    assert rel(dcond, 0.5 * cr.grad) < 1e-5 and rel(dwl, wr.grad) < 1e-5 and rel(dbl, br.grad) < 1e-5


def test_composition_fwd_bwd(ops):
    g = torch.Generator().manual_seed(2)
    B, H, W = 2, 9, 7
    z0 = torch.randn(B, 2, H, W, generator=g)
    z1 = torch.randn(B, 5, H, W, generator=g)
    z0r, z1r = z0.clone().requires_grad_(True), z1.clone().requires_grad_(True)
    p0 = torch.sigmoid(z0r)
    parts, start = [], 0
    for parent, size in ((1, 2), (0, 3)):
        pp = p0[:, parent:parent + 1]
        q = torch.softmax(z1r[:, start:start + size] + torch.log(pp + 1e-6), 1)
        parts.append(pp * q)
        start += size
    p1 = torch.cat(parts, 1)
    d1 = torch.randn(p1.shape, generator=g)
    p1.backward(d1)

    p0d = ops.sigmoid_fwd(z0.cuda())
    p1d = ops.compose_fwd(z1.cuda(), p0d, [1, 0], [2, 3])
    assert rel(p0d, p0) < 1e-6 and rel(p1d, p1) < 1e-5
    dz1, dp0 = ops.compose_bwd(d1.cuda(), z1.cuda(), p0d, [1, 0], [2, 3])
    dz0 = ops.sigmoid_bwd(dp0, z0.cuda())
    assert rel(dz1, z1r.grad) < 1e-5 and rel(dz0, z0r.grad) < 1e-5
    # broadcast (GAP-style) gradient: [B,C] expanded over pixels, no materialisation
    dc = torch.randn(B, 5, generator=g)
    z1r.grad = None
    z0r.grad = None
    p0 = torch.sigmoid(z0r)
    parts, start = [], 0
    for parent, size in ((1, 2), (0, 3)):
        pp = p0[:, parent:parent + 1]
        parts.append(pp * torch.softmax(z1r[:, start:start + size] + torch.log(pp + 1e-6), 1))
        start += size
    (torch.cat(parts, 1) * dc[:, :, None, None]).sum().backward()
    dz1b, dp0b = ops.compose_bwd(dc.cuda()[:, :, None, None].expand(B, 5, H, W), z1.cuda(), p0d, [1, 0], [2, 3])
    assert rel(dz1b, z1r.grad) < 1e-5
    assert rel(ops.sigmoid_bwd(dp0b, z0.cuda()), z0r.grad) < 1e-5


def test_loss_against_golden_and_oracle(ops):
    from oracle import losses as OL
    from tests.helpers import load_golden
    gold = load_golden("loss_cases")
    w = [float(v) for v in gold["w"]]
    z, t = torch.from_numpy(gold["z"]), torch.from_numpy(gold["t"])
    out, coef = ops.loss_fwd(z.cuda(), t.cuda(), torch.tensor(w).cuda())
    assert abs(float(out[0]) - gold["ce"]) < 2e-6 and abs(float(out[1]) - gold["dice"]) < 2e-6
    dz = ops.loss_bwd(z.cuda(), t.cuda(), coef, torch.ones(2).cuda())
    assert rel(dz, torch.from_numpy(gold["dz"])) < 2e-5
    out2, _ = ops.loss_fwd(torch.from_numpy(gold["z_all"]).cuda(), torch.from_numpy(gold["t_all"]).cuda(),
                           torch.tensor(w).cuda())
    assert abs(float(out2[0]) - 1.0) < 1e-6 and float(out2[2]) == 0.0
    # 7-class flat level, separate upstream gradients
    g = torch.Generator().manual_seed(4)
    z7 = torch.randn(2, 7, 33, 31, generator=g)
    t7 = F.one_hot(torch.randint(0, 7, (2, 33, 31), generator=g), 7).permute(0, 3, 1, 2).float()
    w7 = [0.0285, 1.5159, 0.9227, 1.4842, 0.2532, 1.0, 3.8021]
    zr = z7.clone().requires_grad_(True)
    ce, dice = OL.cross_entropy_loss(zr, t7, True, w7), OL.soft_dice_loss(zr, t7, True, w7)
    (0.7 * ce + 1.3 * dice).backward()
    out7, coef7 = ops.loss_fwd(z7.cuda(), t7.cuda(), torch.tensor(w7).cuda())
    assert abs(float(out7[0]) - ce.item()) < 2e-6 and abs(float(out7[1]) - dice.item()) < 2e-6
    dz7 = ops.loss_bwd(z7.cuda(), t7.cuda(), coef7, torch.tensor([0.7, 1.3]).cuda())
    assert rel(dz7, zr.grad) < 2e-5


def test_consistency_and_metrics(ops):
    from oracle import metrics as OM
    g = torch.Generator().manual_seed(6)
    B, H, W = 2, 21, 19
    z0, z1 = torch.randn(B, 4, H, W, generator=g), torch.randn(B, 4, H, W, generator=g)
    lab = torch.randint(0, 7, (B, H, W), generator=g)
    t0 = torch.stack([lab == 0, lab == 1, lab == 2, lab >= 3], 1).float()
    t1 = torch.stack([lab == 3, lab == 4, lab == 5, lab == 6], 1).float()
    t1 = torch.where((lab < 3)[:, None], torch.full_like(t1, -1.0), t1)
    oh0, cm0 = ops.predict_metrics(z0.cuda(), t0.cuda(), child=False)
    oh1, cm1 = ops.predict_metrics(z1.cuda(), t1.cuda(), child=True)
    ref = OM.train_step_metrics([z0.numpy(), z1.numpy()], [t0.numpy(), t1.numpy()])
    from hrseg_amd.Metrics.performance_metrics import metrics_from_confusion
    for L, (cm, child) in enumerate(((cm0, False), (cm1, True))):
        got = metrics_from_confusion(cm, child)
        for k in OM.METRIC_NAMES:
            assert np.allclose(got[k].cpu().numpy(), ref[k][4 * L:4 * L + 4], atol=1e-6), (L, k)
    oh_ref = [np.where(t.numpy() == -1, 0.0, OM.one_hot_predictions(z.numpy())) for z, t in ((z0, t0), (z1, t1))]
    assert np.array_equal(oh0.cpu().numpy(), oh_ref[0]) and np.array_equal(oh1.cpu().numpy(), oh_ref[1])
    sums = ops.consistency_sums(oh1, oh0, [3], [4])
    ref_c = np.abs(oh_ref[1].sum(1) - oh_ref[0][:, 3]).sum()
    assert abs(float(sums[0]) - ref_c) < 1e-6


def test_adamw_matches_torch(ops):
    g = torch.Generator().manual_seed(8)
    n = 10007
    p = torch.randn(n, generator=g)
    pr = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([pr], lr=1e-3)
    pd, m, v = p.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone()
        opt.step()
        ops.adamw(pd, gr.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
    assert rel(pd, pr.data) < 1e-6


def test_grouped_conv_matches_single_launches(ops):
    """four independent convs (HRNet branch shapes) through the grouped entry points"""
    g = torch.Generator().manual_seed(11)
    chans, sizes = [48, 96, 192, 384], [33, 17, 9, 5]
    xs = [torch.randn(2, h, h, c, generator=g).cuda() for c, h in zip(chans, sizes)]
    ws = [(torch.randn(c, 9, c, generator=g) / (9 * c) ** 0.5).cuda() for c in chans]
    ys = ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans)
    dys = [torch.randn(y.shape, generator=g).cuda() for y in ys]
    wts = [ops.weight_transpose(w, c, 9, c) for w, c in zip(ws, chans)]
    dws = [torch.zeros_like(w) for w in ws]
    ops.conv_wgrad_group(xs, dys, dws, 3, 1)
    seed = [torch.randn(x.shape, generator=g).cuda() for x in xs]
    dxs = ops.conv_dgrad_group(dys, wts, [x.shape for x in xs], 3, 1, [None, seed[1].clone(), None, seed[3].clone()],
                               [False, True, False, True])
    for i in range(4):
        y1 = ops.conv_fwd(xs[i], ws[i], None, 3, 1)
        assert rel(ys[i], y1) < 1e-5
        dw1 = torch.zeros_like(ws[i])
        ops.conv_wgrad(xs[i], dys[i], dw1, 3, 1)
        assert rel(dws[i], dw1) < 2e-5
        dx1 = ops.conv_dgrad(dys[i], wts[i], xs[i].shape, 3, 1)
        if i in (1, 3):
            dx1 = dx1 + seed[i]
        assert rel(dxs[i], dx1) < 1e-5


def test_rccl_comm_single_rank():
    """the library's own RCCL wrappers (hrseg_comm_*): real communicator on one rank -- the in-place sum over
    one rank leaves the bucket unchanged, is stream-ordered, and GradSync(backend="rccl") drives it per bucket"""
    import types
    from hrseg_amd.parallel import GradSync, RcclComm
    comm = RcclComm(0, 1, "cuda:0")
    try:
        t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
        want = t.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        comm.all_reduce_(t, s)
        comm.wait(s, torch.cuda.current_stream())
        assert torch.equal(t, want)
        names = ["stem.0.weight", "layer1.0.w", "transition1.0.w", "stage2.0.w", "shared_head.0.weight"]
        sizes = [4096, 1 << 20, 1 << 19, 1 << 21, 1 << 18]
        slots, off = {}, 0
        for n, sz in zip(names, sizes):
            slots[n] = (off, sz)
            off += sz
        flat = types.SimpleNamespace(slots=slots, numel=off, grad=torch.randn(off, device="cuda"), data=torch.randn(off, device="cuda"))
        model = types.SimpleNamespace(_flat=flat, _grad_hook=None)
        import os
        os.environ["HRSEG_FORCE_SYNC"] = "1"
        try:
            sync = GradSync(model, backend="rccl", comm=comm, min_bucket=1000)
        finally:
            del os.environ["HRSEG_FORCE_SYNC"]
        before = flat.grad.clone()
        for mark in ("shared_head", "transition1", "layer1", "end"):
            sync(mark)
        torch.cuda.synchronize()
        assert torch.equal(flat.grad, before) and sync.launched[0][1] == off and sync.launched[-1][0] == 0
    finally:
        comm.close()


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_conv_groups(ops, seed):
    """seeded random groups of 2..8 independent convolutions (mixed channel counts and image sizes, 1x1 / 3x3,
    stride 1 / 2, accumulate flags, batch 1..3) through the grouped entry points against autograd"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(2, 9))
    k = int(rng.choice([1, 3]))
    s = int(rng.choice([1, 2])) if k == 3 else 1
    base = int(rng.choice([48, 64]))                 # a group shares its channel tiling (48- or 64-multiples)
    g = torch.Generator().manual_seed(seed)
    xs, ws, dys, refs = [], [], [], []
    for _ in range(n):
        cin, cout = base * int(rng.integers(1, 4)), base * int(rng.integers(1, 4))
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(3, 34)), int(rng.integers(3, 34))
        x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_(True)
        y = F.conv2d(x, w, stride=s, padding=(k - 1) // 2)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        xs.append(x), ws.append(w), dys.append(dy), refs.append(y.detach())
    xd, wd, dyd = [nhwc(x.detach()) for x in xs], [store(w.detach()) for w in ws], [nhwc(d) for d in dys]
    couts = [w.shape[0] for w in ws]
    ys = ops.conv_fwd_group(xd, wd, [None] * n, k, s, couts)
    for y, r in zip(ys, refs):
        assert rel(nchw(y), r) < 3e-5
    dws = [torch.zeros_like(w) for w in wd]
    ops.conv_wgrad_group(xd, dyd, dws, k, s)
    for dw, w in zip(dws, ws):
        co, ci = w.shape[0], w.shape[1]
        assert rel(dw.view(co, k, k, ci).permute(0, 3, 1, 2).cpu(), w.grad) < 1e-4
    wts = [ops.weight_transpose(w_, w.shape[0], k * k, w.shape[1]) for w_, w in zip(wd, ws)]
    seeds = [torch.randn(x.shape, generator=g) if rng.random() < 0.5 else None for x in xd]
    outs = [sd.clone().cuda() if sd is not None else None for sd in seeds]   # NHWC seeds the launch adds into
    dxs = ops.conv_dgrad_group(dyd, wts, [x.shape for x in xd], k, s, outs, [o is not None for o in outs])
    for dx, x, sd in zip(dxs, xs, seeds):
        want = x.grad if sd is None else x.grad + sd.permute(0, 3, 1, 2)
        assert rel(nchw(dx), want) < 3e-5


@pytest.mark.parametrize("C,H,W,B,L,res", [(48, 13, 11, 2, 2, False), (96, 7, 9, 1, 4, True), (384, 3, 5, 2, 3, False),
                                           (48, 155, 155, 4, 2, True)])
def test_batchnorm_batched_passes(ops, C, H, W, B, L, res):
    """the BN bookkeeping of the batched level passes: the tensor holds L identical copies of one pass's images
    (stat_div), the running statistics take L updates (repeat), the backward normalises each copy with its own
    gradient (nseg) -- against torch BatchNorm applied to ONE copy L times"""
    g = torch.Generator().manual_seed(C + L)
    y0 = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    r0 = torch.randn(B, C, H, W, generator=g) if res else None
    dzs = [torch.randn(B, C, H, W, generator=g) for _ in range(L)]
    rm_ref, rv_ref = torch.zeros(C), torch.ones(C)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z_refs, dy_refs, dres_refs = [], [], []
    for l in range(L):
        yr = y0.clone().requires_grad_(True)
        rr = r0.clone().requires_grad_(True) if res else None
        z = F.batch_norm(yr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
        z = F.relu(z + rr if res else z)
        z.backward(dzs[l])
        z_refs.append(z.detach()), dy_refs.append(yr.grad), dres_refs.append(rr.grad if res else None)
    y = nhwc(torch.cat([y0] * L))
    it = dict(y=y, gamma=gamma.cuda(), beta=beta.cuda(), rm=torch.zeros(C, device="cuda"), rv=torch.ones(C, device="cuda"),
              nbt=torch.zeros((), dtype=torch.int64, device="cuda"), momentum=0.1, eps=1e-5,
              residual=nhwc(torch.cat([r0] * L)) if res else None, relu=True, repeat=L, stat_div=L)
    (z, coef), = ops.bn_fwd_group([it], True)
    assert rel(nchw(z), torch.cat(z_refs)) < 1e-5
    assert rel(it["rm"], rm_ref) < 1e-5 and rel(it["rv"], rv_ref) < 1e-5 and int(it["nbt"]) == L
    bw = dict(dz=nhwc(torch.cat(dzs)), z=z if res else None, relu=True, y=y, coef=coef, dgamma=torch.zeros(C, device="cuda"),
              dbeta=torch.zeros(C, device="cuda"), dres=torch.empty_like(y) if res else None, dres_accumulate=False, nseg=L)
    dy, = ops.bn_bwd_group([bw], False)
    want = torch.cat(dy_refs)
    assert float((nchw(dy) - want).abs().max()) < 2e-5 * max(float(want.abs().max()), 1.0)
    assert rel(bw["dgamma"], gr.grad) < 2e-5 and rel(bw["dbeta"], br.grad) < 2e-5
    if res:
        assert rel(nchw(bw["dres"]), torch.cat(dres_refs)) < 1e-6


@pytest.mark.parametrize("tree_groups", [
    ([("tooth", ["pulp", "dentin", "enamel", "composite"])], ["background", "upper", "lower", "tooth"]),
    ([("a", ["c", "d"]), ("b", ["e", "f", "g"])], ["a", "b"]),
])
def test_grouped_conditional_kl_fwd_bwd(tree_groups):
    """hrseg_group_kl / hrseg_group_kl_bwd (the opt-in stabiliser of the reference's Metrics/losses.py:180-210) against
    the oracle twin's autograd; logits include near-saturated pixels (the clamp_min(1e-8) branch)"""
    from oracle import losses as OL
    from hrseg_amd.Metrics import losses as PL
    groups, levels_prev = tree_groups
    C = sum(len(ch) for _, ch in groups)
    g = torch.Generator().manual_seed(C)
    z = torch.randn(2, C, 33, 29, generator=g) * 3.0
    z[0, 0, :4, :4] = 40.0                                   # saturated softmax
    pp = torch.rand(2, len(levels_prev), 33, 29, generator=g)
    zo = z.clone().requires_grad_(True)
    lo = OL.grouped_conditional_kl(zo, pp, groups, levels_prev)
    lo.backward()
    zp = z.cuda().requires_grad_(True)
    ppd = pp.cuda().requires_grad_(True)
    lp = PL.grouped_conditional_kl(zp, ppd, groups, levels_prev)
    assert abs(float(lp) - float(lo)) < 1e-5 * max(1.0, abs(float(lo)))
    (3.0 * lp).backward()
    assert float((zp.grad.cpu() / 3.0 - zo.grad).abs().max()) < 1e-5 * float(zo.grad.abs().max()) + 1e-9
    assert ppd.grad is None                                   # constant log-bias inside a group: no gradient


def test_get_loss_adds_the_kl_term_only_when_asked():
    import argparse
    from oracle import losses as OL
    from hrseg_amd import train as PT
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd.utils.hierarchy import build_hierarchy_indices, child_groups
    from tests.helpers import load_tree, level_weights_for
    tree = load_tree("class_tree_tl.json")
    levels, parent_of, children_of = build_hierarchy_indices(tree)
    model = argparse.Namespace(levels=levels, parent_of=parent_of, child_groups=child_groups(levels, children_of))
    g = torch.Generator().manual_seed(3)
    logits = [torch.randn(2, 4, 16, 16, generator=g).cuda() for _ in range(2)]
    lab = torch.randint(0, 4, (2, 16, 16), generator=g)
    targets = [torch.nn.functional.one_hot(lab, 4).permute(0, 3, 1, 2).float().cuda() for _ in range(2)]
    fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=4)] for _ in range(2)]
    w = level_weights_for("class_tree_tl.json", True)
    probs = [torch.sigmoid(z) for z in logits]
    base, _, _ = PT.get_loss(logits, targets, fns, [], w, 0.0, [], probs_per_level=probs, model=model)
    with_kl, _, _ = PT.get_loss(logits, targets, fns, [], w, 0.0, [], probs_per_level=probs, model=model, lambda_kl=0.1)
    want = 0.1 * float(OL.grouped_conditional_kl(logits[1].cpu(), probs[0].cpu(), model.child_groups[0], levels[0]))
    assert abs(float(with_kl) - float(base) - want) < 1e-5


@pytest.mark.parametrize("cin,stride,bias", [(7, 1, True), (7, 2, False), (5, 2, False), (3, 1, True)])
def test_first_layer_direct_conv_kernels(cin, stride, bias):
    """the direct (VALU) kernels of the first layer for Cin <= 8 -- image + previous level's logits with logit-concatenated
    re-encoding -- forward, weight gradient and the data gradient (which reads the FORWARD weight layout) vs torch"""
    import torch.nn.functional as F
    from hrseg_amd import ops
    g = torch.Generator().manual_seed(cin * 10 + stride)
    B, H, W, cout = 3, 37, 45, 64
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, b, stride=stride, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()                  # [Cout][3][3][Cin]
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    yd = ops.conv_fwd(xd, wd.reshape(cout, 9, cin), b.cuda() if bias else None, 3, stride)
    assert float((yd.permute(0, 3, 1, 2).cpu() - y.detach()).abs().max()) < 1e-5 * float(y.abs().max())
    dw = torch.zeros_like(wd)
    ops.conv_wgrad(xd, dyd, dw, 3, stride)
    assert float((dw.permute(0, 3, 1, 2).cpu() - wr.grad).abs().max()) < 2e-5 * float(wr.grad.abs().max())
    dx = ops.conv_dgrad(dyd, wd.reshape(cout, 9, cin), xd.shape, 3, stride)
    assert float((dx.permute(0, 3, 1, 2).cpu() - xr.grad).abs().max()) < 1e-5 * float(xr.grad.abs().max())
    ops.conv_dgrad(dyd, wd.reshape(cout, 9, cin), xd.shape, 3, stride, out=dx, accumulate=True)
    assert float((dx.permute(0, 3, 1, 2).cpu() - 2 * xr.grad).abs().max()) < 1e-5 * float(xr.grad.abs().max())


def test_bn_backward_with_relu_mask_bytes_equals_reading_z():
    """layers with a residual: the forward leaves its ReLU mask as one byte per channel quad; the backward that reads those
    bytes must give the bits of the backward that reads z"""
    from hrseg_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W, C = 3, 21, 17, 48
    y = torch.randn(B, H, W, C, generator=g).cuda()
    res = torch.randn(B, H, W, C, generator=g).cuda()
    dz = torch.randn(B, H, W, C, generator=g).cuda()
    mask = torch.empty((B * H * W, C // 4), dtype=torch.uint8, device="cuda")

    def item(**kw):
        return dict(y=y, gamma=torch.ones(C, device="cuda") * 1.3, beta=torch.zeros(C, device="cuda") - 0.1,
                    rm=torch.zeros(C, device="cuda"), rv=torch.ones(C, device="cuda"),
                    nbt=torch.zeros((), dtype=torch.int64, device="cuda"), momentum=0.1, eps=1e-5, residual=res, relu=True, **kw)
    (z, coef), = ops.bn_fwd_group([item(relu_mask=mask)], True)
    want_bits = (z > 0).view(B * H * W, C // 4, 4).to(torch.uint8)
    want = want_bits[..., 0] | (want_bits[..., 1] << 1) | (want_bits[..., 2] << 2) | (want_bits[..., 3] << 3)
    assert torch.equal(mask, want)
    outs = []
    for use_mask in (True, False):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dres = torch.empty_like(y)
        d = dz.clone()
        ops.bn_bwd_group([dict(dz=d, z=None if use_mask else z, relu_mask=mask if use_mask else None, relu=True, y=y, coef=coef,
                               dgamma=dg, dbeta=db, dres=dres, dres_accumulate=False)], False)
        outs.append((d, dg, db, dres))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n_same,n_low", [(1, 3), (2, 2), (3, 1), (4, 0), (1, 1)])
def test_fuse_sum_one_pass_equals_the_launch_chain(n_same, n_low):
    """hrseg_fuse_sum (HRNet fuse layer sum, models.py:527-542: same-resolution terms + up-sampled low-resolution terms +
    ReLU in one pass) adds in the order of the add / accumulating-bilinear chain it replaces: identical bits, and equal to
    torch's F.interpolate(align_corners=True) sum within fp32 rounding"""
    import torch.nn.functional as F
    from hrseg_amd import ops
    g = torch.Generator().manual_seed(10 * n_same + n_low)
    B, H, W, C = 2, 39, 41, 48
    same = [torch.randn(B, H, W, C, generator=g).cuda() for _ in range(n_same)]
    lows = [torch.randn(B, (H + 2 ** (j + 1) - 1) // 2 ** (j + 1), (W + 2 ** (j + 1) - 1) // 2 ** (j + 1), C, generator=g).cuda()
            for j in range(n_low)]
    got = ops.fuse_sum(same, lows, relu=True, align_corners=True)
    # the launch chain of engine.fuse_sum before the one-pass kernel
    if n_same >= 2:
        out = ops.add(same[0], same[1], relu=(n_same == 2 and not lows))
        for i, a in enumerate(same[2:]):
            ops.add(out, a, relu=(i == n_same - 3 and not lows), out=out)
    else:
        out = same[0].clone()
    for i, a in enumerate(lows):
        ops.bilinear_fwd(a, out, H, W, 0, 0, True, accumulate=True, relu=(i == n_low - 1))
    if n_same == 1 and not lows:
        out = torch.relu(out)
    assert torch.equal(got, out)
    ref = sum(t.permute(0, 3, 1, 2).cpu() for t in same)
    for a in lows:
        ref = ref + F.interpolate(a.permute(0, 3, 1, 2).cpu(), size=(H, W), mode="bilinear", align_corners=True)
    ref = torch.relu(ref)
    assert float((got.permute(0, 3, 1, 2).cpu() - ref).abs().max()) < 1e-5 * float(ref.abs().max())


def test_metric_vectors_kernel_equals_the_per_level_tensor_arithmetic(ops):
    """hrseg_metric_vectors (one launch for all levels) against metrics_from_confusion, the torch-op form the metric classes
    use and the oracle pins: classes nobody predicted, classes absent from the targets, an all-background child level"""
    from hrseg_amd.Metrics.performance_metrics import METRIC_NAMES, metrics_from_confusion
    g = torch.Generator().manual_seed(11)
    cms = [torch.randint(0, 50000, (4, 4), generator=g), torch.randint(0, 50000, (5, 5), generator=g),
           torch.randint(0, 9, (8, 8), generator=g), torch.zeros(3, 3, dtype=torch.int64)]
    cms[0][:, 2] = 0                # class 2 never predicted
    cms[1][3, :] = 0                # class 3 absent from the targets
    cms[1][:, 3] = 0                # ... and never predicted: every denominator of that class is zero
    cms[3][0, 0] = 1234             # a child level that saw background only
    cms = [c.cuda() for c in cms]
    child = [False, True, True, True]
    vec = ops.metric_vectors(cms, child)
    want = {k: [] for k in METRIC_NAMES}
    for cm, ch in zip(cms, child):
        m = metrics_from_confusion(cm, child_classes=ch)
        for k in METRIC_NAMES:
            want[k].append(m[k])
    assert vec.shape == (5, 4 + 4 + 7 + 2)
    for i, k in enumerate(METRIC_NAMES):
        assert torch.equal(vec[i], torch.cat(want[k])), k


@pytest.mark.parametrize("stride,H,W,B,cout", [(2, 61, 77, 3, 64), (1, 33, 46, 2, 64), (2, 300, 301, 2, 64), (1, 19, 20, 2, 128)])
def test_three_channel_first_layer_kernels_equal_the_generic_ones(ops, stride, H, W, B, cout):
    """the stem's 3 -> 64 3x3 convolution has kernels of its own (weights in registers, 16-byte stores; 27 sums per thread in
    the weight gradient): the forward -- bias, residual and ReLU included -- must equal the generic Cin <= 8 kernel bit for bit
    (same accumulation order; hrseg_tune small_cin3=0 runs the generic one), the weight gradient agrees to fp32 summation
    order, and both match torch"""
    from hrseg_amd import _lib
    g = torch.Generator().manual_seed(H + W + stride)
    x = torch.randn(B, 3, H, W, generator=g)
    w = (torch.randn(cout, 3, 3, 3, generator=g) / 5).requires_grad_(True)
    bias = torch.randn(cout, generator=g)
    y_ref = F.conv2d(x, w, bias, stride=stride, padding=1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    res = torch.randn(y_ref.shape, generator=g)
    xd, wd, dyd, rd = nhwc(x), store(w.detach()), nhwc(dy), nhwc(res)
    out = {}
    try:
        for new in (1, 0):
            _lib.tune(small_cin3=new)
            y = ops.conv_fwd(xd, wd, bias.cuda(), 3, stride)
            yr = ops.conv_fwd(xd, wd, bias.cuda(), 3, stride, residual=rd, relu=True)
            dw = torch.zeros_like(wd)
            ops.conv_wgrad(xd, dyd, dw, 3, stride)
            ops.conv_wgrad(xd, dyd, dw, 3, stride)
            out[new] = (y, yr, dw)
    finally:
        _lib.tune(small_cin3=1)
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1])
    assert rel(nchw(out[1][0]), y_ref) < 1e-5
    assert rel(nchw(out[1][1]), torch.relu(y_ref.detach() + res)) < 1e-5
    assert rel(out[1][2], out[0][2]) < 2e-5
    assert rel(out[1][2].view(cout, 3, 3, 3).permute(0, 3, 1, 2).cpu(), 2 * w.grad) < 5e-5
