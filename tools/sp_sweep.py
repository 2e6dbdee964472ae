"""Tile-plan sweep of the fp16x2 im2col kernel on the low-resolution branch convolutions.
    python tools/sp_sweep.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [("192@39", 192, 39), ("384@20", 384, 20), ("96@78", 96, 78)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, c, h in SHAPES:
    x = torch.randn(B, h, h, c, device="cuda")
    w = torch.randn(c, 9, c, device="cuda") * 0.05
    fl = 2.0 * B * h * h * c * c * 9
    y = ops.conv_fwd(x, w, None, 3, 1, prec=0)
    t = timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, out=y, prec=0))
    print("%s f32 auto-plan: %.1f us %.0f TF" % (name, t, fl / t / 1e6), flush=True)
    for patch in (0, 1):
        for wtm in (1, 2, 4):
            for wtn in (3, 6):
                for ks in (1, 2, 3, 4, 6):
                    if patch and (wtm != 1 or ks != 1):
                        continue
                    _lib.tune(sp_patch=patch, sp_wtm=wtm, sp_wtn=wtn, sp_ksplit=ks)
                    try:
                        t = timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, out=y, prec=5))
                    except Exception as e:
                        print("  skip", patch, wtm, wtn, ks, str(e)[:60])
                        continue
                    print("%s fp16x2 patch=%d wtm=%d wtn=%d ksplit=%d: %.1f us %.0f TF" % (name, patch, wtm, wtn, ks, t, fl / t / 1e6), flush=True)
    _lib.tune(sp_patch=1, sp_wtm=0, sp_wtn=0, sp_ksplit=0)
