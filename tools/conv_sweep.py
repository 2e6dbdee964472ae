"""Sweep the implicit-GEMM tile plan (pixel tiles/wave, K chunks, LDS buffers, split-K) per layer shape
(forward kernel; dgrad is the same kernel).  Prints the best plans per shape."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops, _lib

PEAK = 157.3
B = int(os.environ.get("SWEEP_B", "4"))
SHAPES = [
    ("hr 48->48 3x3 @155", 48, 48, 3, 1, 155), ("hr 96->96 3x3 @78", 96, 96, 3, 1, 78),
    ("hr 192->192 3x3 @39", 192, 192, 3, 1, 39), ("hr 384->384 3x3 @20", 384, 384, 3, 1, 20),
    ("hr 720->720 1x1 @155", 720, 720, 1, 1, 155), ("hr 64->64 3x3 @155", 64, 64, 3, 1, 155),
    ("hr 64->256 1x1 @155", 64, 256, 1, 1, 155), ("hr 256->64 1x1 @155", 256, 64, 1, 1, 155),
    ("hr 48->96 3x3s2 @155", 48, 96, 3, 2, 155), ("hr 96->192 3x3s2 @78", 96, 192, 3, 2, 78),
    ("hr 384->48 1x1 @20", 384, 48, 1, 1, 20), ("hr 192->48 1x1 @39", 192, 48, 1, 1, 39),
    ("hr 96->48 1x1 @78", 96, 48, 1, 1, 78),
    ("hr 256->48 3x3 @155", 256, 48, 3, 1, 155), ("hr 256->96 3x3s2 @155", 256, 96, 3, 2, 155),
    ("hr 64->64 1x1 @155", 64, 64, 1, 1, 155), ("hr 64->64 3x3s2 @310", 64, 64, 3, 2, 310),
    ("hr 48->48 3x3s2 @155", 48, 48, 3, 2, 155), ("hr 48->384 3x3s2 @39", 48, 384, 3, 2, 39),
    ("un 64->64 3x3 @620", 64, 64, 3, 1, 620), ("un 256->256 3x3 @155", 256, 256, 3, 1, 155),
    ("un 512->512 3x3 @38", 512, 512, 3, 1, 38), ("un 1024->256 3x3 @77", 1024, 256, 3, 1, 77),
]
only = sys.argv[1] if len(sys.argv) > 1 else None


def timeit(fn, n=8):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for name, ci, co, k, s, H in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(B, H, H, ci, device="cuda")
    w = torch.randn(co, k * k, ci, device="cuda") * 0.05
    _lib.set_conv_tune()
    y = ops.conv_fwd(x, w, None, k, s)
    flops = 2.0 * y.numel() * ci * k * k
    t_auto = timeit(lambda: ops.conv_fwd(x, w, None, k, s, out=y))
    res = []
    for wtm, kc, db, ks in itertools.product((1, 2, 4), (1, 2, 3), (1, 2), (1, 2, 3, 4, 6)):
        if ci % (16 * kc):
            continue
        if ks > 1 and y.numel() * 4 > 64e6:
            continue
        wtn = 3 if co % 48 == 0 else 4
        lds = (64 * wtm + 16 * wtn) * 16 * kc * 4 * db
        if lds > 160 * 1024:
            continue
        _lib.set_conv_tune(wtm, kc, db, ks)
        try:
            t = timeit(lambda: ops.conv_fwd(x, w, None, k, s, out=y))
        except RuntimeError as e:
            continue
        res.append((t, wtm, kc, db, ks, lds))
    _lib.set_conv_tune()
    res.sort()
    best = ", ".join("wtm%d kc%d db%d ks%d %.1fus(%.0f%%)" % (r[1], r[2], r[3], r[4], r[0] * 1e6, 100 * flops / r[0] / 1e12 / PEAK)
                     for r in res[:5])
    print("%-24s auto %.1fus (%.0f%%) | %s" % (name, t_auto * 1e6, 100 * flops / t_auto / 1e12 / PEAK, best), flush=True)
