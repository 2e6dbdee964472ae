"""Split-precision conv kernels (bf16x3 / bf16x2 / bf16) against the exact-fp32 MFMA kernels: error and time
per shape, forward / data gradient / weight gradient, single launches and the grouped branch launch.
    python tools/sp_bench.py [B] [name filter]
Error = max |sp - f32| / max |f32| (the f32 kernel itself is ~1e-6 from an fp64 evaluation)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import _lib, ops

SHAPES = [  # name, Cin, Cout, k, s, H, W
    ("hr 48->48 3x3 @155", 48, 48, 3, 1, 155, 155), ("hr 96->96 3x3 @78", 96, 96, 3, 1, 78, 78),
    ("hr 192->192 3x3 @39", 192, 192, 3, 1, 39, 39), ("hr 384->384 3x3 @20", 384, 384, 3, 1, 20, 20),
    ("hr 720->720 1x1 @155", 720, 720, 1, 1, 155, 155), ("hr 64->64 3x3 @155", 64, 64, 3, 1, 155, 155),
    ("hr 64->256 1x1 @155", 64, 256, 1, 1, 155, 155), ("hr 256->64 1x1 @155", 256, 64, 1, 1, 155, 155),
    ("hr 48->96 3x3s2 @155", 48, 96, 3, 2, 155, 155), ("hr 384->48 1x1 @20", 384, 48, 1, 1, 20, 20),
    ("hr 96->48 1x1 @78", 96, 48, 1, 1, 78, 78),
    ("un 64->64 3x3 @620", 64, 64, 3, 1, 620, 620), ("un 256->256 3x3 @155", 256, 256, 3, 1, 155, 155),
]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
only = sys.argv[2] if len(sys.argv) > 2 else None
MODES = [m for m in os.environ.get("SP_MODES", "f32,bf16x3,fp16x2,bf16x2,bf16").split(",")]
DYSCALE = float(os.environ.get("SP_DYSCALE", "1"))       # magnitude of the gradient operand (fp16x2 scales by its |max|)
WHAT = os.environ.get("SP_WHAT", "fwd,dgrad,wgrad").split(",")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def err(a, b):
    return float((a - b).abs().max() / b.abs().max())


for name, ci, co, k, s, H, W in SHAPES:
    if only and only not in name:
        continue
    if B * H * W * max(ci, co) * 4 > 3e9:
        continue
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, H, W, ci, device="cuda", generator=g)
    w = torch.randn(co, k * k, ci, device="cuda", generator=g) * 0.05
    y0 = ops.conv_fwd(x, w, None, k, s)
    dy = torch.randn(y0.shape, device="cuda", generator=g) * DYSCALE
    gmax_t = dy.abs().max().reshape(1).repeat(64)
    wt = ops.weight_transpose(w, co, k * k, ci)
    flops = 2.0 * y0.numel() * ci * k * k
    ref = {}
    line = "%-22s %6.1f GF |" % (name, flops / 1e9)
    for mode in MODES:
        pr = _lib.CONV_PRECISION[mode]
        gm = gmax_t if mode in ("fp16x2", "auto") else None
        out = []
        if "fwd" in WHAT:
            y = ops.conv_fwd(x, w, None, k, s, prec=pr)
            t = timeit(lambda: ops.conv_fwd(x, w, None, k, s, out=y, prec=pr))
            ref.setdefault("fwd", y.clone())
            out.append("fwd %6.1f us %5.0f TF e=%.1e" % (t * 1e6, flops / t / 1e12, err(y, ref["fwd"])))
        if "dgrad" in WHAT:
            dx = ops.conv_dgrad(dy, wt, x.shape, k, s, prec=pr, gmax=gm)
            t = timeit(lambda: ops.conv_dgrad(dy, wt, x.shape, k, s, out=dx, prec=pr, gmax=gm))
            ref.setdefault("dgrad", dx.clone())
            out.append("dgrad %6.1f us %5.0f TF e=%.1e" % (t * 1e6, flops / t / 1e12, err(dx, ref["dgrad"])))
        if "wgrad" in WHAT:
            dw = torch.zeros_like(w)
            ops.conv_wgrad(x, dy, dw, k, s, prec=pr, gmax=gm)
            dw1 = dw.clone()
            t = timeit(lambda: ops.conv_wgrad(x, dy, dw, k, s, prec=pr, gmax=gm))
            ref.setdefault("wgrad", dw1)
            out.append("wgrad %6.1f us %5.0f TF e=%.1e" % (t * 1e6, flops / t / 1e12, err(dw1, ref["wgrad"])))
        print(line + " %-7s " % mode + " | ".join(out), flush=True)

if not only or "group" in only:
    sizes, chans = [155, 78, 39, 20], [48, 96, 192, 384]
    xs = [torch.randn(B, h, h, c, device="cuda") for c, h in zip(chans, sizes)]
    ws = [torch.randn(c, 9, c, device="cuda") * 0.05 for c in chans]
    dys = [torch.randn(B, h, h, c, device="cuda") for c, h in zip(chans, sizes)]
    fl = [2.0 * B * h * h * c * c * 9 for c, h in zip(chans, sizes)]
    for n in (2, 3, 4):
        base = None
        for mode in MODES:
            pr = _lib.CONV_PRECISION[mode]
            ys = ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n], prec=pr)
            base = base or [y.clone() for y in ys]
            e = max(err(a, b) for a, b in zip(ys, base))
            t = timeit(lambda: ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n], prec=pr))
            dws = [torch.zeros_like(w) for w in ws[:n]]
            gms = [d.abs().max().reshape(1).repeat(64) for d in dys[:n]] if mode in ("fp16x2", "auto") else None
            tw = timeit(lambda: ops.conv_wgrad_group(xs[:n], dys[:n], dws, 3, 1, prec=pr, gmaxs=gms))
            print("group of %d branch convs  %-7s fwd %7.1f us %5.0f TF e=%.1e | wgrad %7.1f us %5.0f TF" % (
                n, mode, t * 1e6, sum(fl[:n]) / t / 1e12, e, tw * 1e6, sum(fl[:n]) / tw / 1e12), flush=True)
