"""Sweep the wgrad kernel plan (pixels per stage, LDS buffers, target grid) per layer shape."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops, _lib
PEAK = 157.3
B = 4
SHAPES = [
    ("hr 48->48 3x3 @155", 48, 48, 3, 1, 155), ("hr 96->96 3x3 @78", 96, 96, 3, 1, 78),
    ("hr 192->192 3x3 @39", 192, 192, 3, 1, 39), ("hr 384->384 3x3 @20", 384, 384, 3, 1, 20),
    ("hr 720->720 1x1 @155", 720, 720, 1, 1, 155), ("hr 64->64 3x3 @155", 64, 64, 3, 1, 155),
    ("hr 64->256 1x1 @155", 64, 256, 1, 1, 155), ("hr 48->96 3x3s2 @155", 48, 96, 3, 2, 155),
    ("un 64->64 3x3 @620", 64, 64, 3, 1, 620), ("un 256->256 3x3 @155", 256, 256, 3, 1, 155),
    ("un 1024->256 3x3 @77", 1024, 256, 3, 1, 77),
]
def timeit(fn, n=6):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for name, ci, co, k, s, H in SHAPES:
    x = torch.randn(B, H, H, ci, device="cuda")
    w = torch.randn(co, k * k, ci, device="cuda") * 0.05
    y = ops.conv_fwd(x, w, None, k, s)
    dy = torch.randn_like(y); dw = torch.zeros_like(w)
    flops = 2.0 * y.numel() * ci * k * k
    res = []
    for pix, db, tb in itertools.product((64, 128), (1, 2), (512, 1024, 2048, 4096)):
        _lib.set_wgrad_tune(pix, db, tb)
        try:
            t = timeit(lambda: ops.conv_wgrad(x, dy, dw, k, s))
        except RuntimeError:
            continue
        res.append((t, pix, db, tb))
    _lib.set_wgrad_tune()
    res.sort()
    print("%-24s | %s" % (name, ", ".join("pix%d db%d tb%d %.1fus(%.0f%%)" % (r[1], r[2], r[3], r[0] * 1e6, 100 * flops / r[0] / 1e12 / PEAK) for r in res[:5])), flush=True)
