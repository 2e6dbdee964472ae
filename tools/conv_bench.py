"""Per-shape timing of the conv kernels (fwd / dgrad / wgrad) on the HRNet-W48 / UNet layer shapes
at B=4, 620x620.  Events on the launch stream; random data; prints TFLOP/s and fraction of the
fp32 MFMA peak (157.3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops

PEAK = 157.3
SHAPES = [  # name, Cin, Cout, k, s, H, W   (B=4)
    ("hr 48->48 3x3 @155", 48, 48, 3, 1, 155, 155), ("hr 96->96 3x3 @78", 96, 96, 3, 1, 78, 78),
    ("hr 192->192 3x3 @39", 192, 192, 3, 1, 39, 39), ("hr 384->384 3x3 @20", 384, 384, 3, 1, 20, 20),
    ("hr 720->720 1x1 @155", 720, 720, 1, 1, 155, 155), ("hr 64->64 3x3 @155", 64, 64, 3, 1, 155, 155),
    ("hr 64->256 1x1 @155", 64, 256, 1, 1, 155, 155), ("hr 256->64 1x1 @155", 256, 64, 1, 1, 155, 155),
    ("hr 48->96 3x3s2 @155", 48, 96, 3, 2, 155, 155), ("hr 384->48 1x1 @20", 384, 48, 1, 1, 20, 20),
    ("un 64->64 3x3 @620", 64, 64, 3, 1, 620, 620), ("un 128->64 3x3 @620", 128, 64, 3, 1, 620, 620),
    ("un 256->256 3x3 @155", 256, 256, 3, 1, 155, 155), ("un 1024->256 3x3 @77", 1024, 256, 3, 1, 77, 77),
    ("un 512->512 3x3 @38", 512, 512, 3, 1, 38, 38),
]
B = 4
only = sys.argv[1] if len(sys.argv) > 1 else None


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


tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
for name, ci, co, k, s, H, W in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(B, H, W, ci, device="cuda")
    w = torch.randn(co, k * k, ci, device="cuda") * 0.05
    y = ops.conv_fwd(x, w, None, k, s)
    dy = torch.randn_like(y)
    wt = ops.weight_transpose(w, co, k * k, ci)
    dw = torch.zeros_like(w)
    flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * co * ci * k * k
    tf = timeit(lambda: ops.conv_fwd(x, w, None, k, s, out=y))
    td = timeit(lambda: ops.conv_dgrad(dy, wt, x.shape, k, s, out=x))
    tw = timeit(lambda: ops.conv_wgrad(x, dy, dw, k, s))
    for key, t in (("fwd", tf), ("dgrad", td), ("wgrad", tw)):
        tot[key][0] += flops
        tot[key][1] += t
    print("%-24s %7.2f GF | fwd %7.1f us %5.1f TF %4.1f%% | dgrad %7.1f us %5.1f TF %4.1f%% | wgrad %7.1f us %5.1f TF %4.1f%%" % (
        name, flops / 1e9, tf * 1e6, flops / tf / 1e12, 100 * flops / tf / 1e12 / PEAK, td * 1e6, flops / td / 1e12,
        100 * flops / td / 1e12 / PEAK, tw * 1e6, flops / tw / 1e12, 100 * flops / tw / 1e12 / PEAK), flush=True)
for key, (f, t) in tot.items():
    if t:
        print("total %-6s %6.1f TF (%.1f%%)" % (key, f / t / 1e12, 100 * f / t / 1e12 / PEAK))
