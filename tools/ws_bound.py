"""What bounds the wave-specialised 3x3 kernel: BALANCED single-problem forward launches (every persistent block walks exactly
25 tiles), timed per tile, for the four tilings of the headline step.  Run against measurement-only builds of conv_ws.hip
(-DHRSEG_WS_EXP=bits: 1 no MFMAs, 2 no weight loads, 4 no patch loads, 8 no weight stores, 16 no epilogue) to see which part
of a tile's time is exposed.      python tools/ws_bound.py [label]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hrseg_amd import _lib, ops

dev = torch.device("cuda:0")
pr = _lib.CONV_PRECISION["auto"]
label = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (channels, H, B, rows per tile, channel tile) -> 6400 tiles = 25 per block
CASES = [(48, 155, 64, 16, 48), (96, 78, 128, 8, 96), (192, 78, 64, 8, 96), (384, 78, 32, 8, 96), (64, 78, 128, 8, 64)]
g = torch.Generator(device="cuda").manual_seed(1)
row = []
for c, h, b, th, bn in CASES:
    x = torch.randn(b, h, h, c, device=dev, generator=g)
    w = torch.randn(c, 9, c, device=dev, generator=g) * 0.05
    y = torch.empty_like(x)
    tiles = b * ((h + 15) // 16) * ((h + th - 1) // th) * (c // bn)
    t = timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, out=y, prec=pr))
    slabs = (c // (16 * (4 if bn == 64 else 3))) * ((9 * (4 if bn == 64 else 3) + 1) // 2)
    mf = (th // 4) * (bn // 16) * 3 * 16                      # MFMA cycles of a slab per SIMD
    per_tile = t * 256 / tiles
    row.append("%d@%d: %7.1f us  %5.2f us/tile  %4.0f cyc/slab (MFMA %d)" % (c, h, t, per_tile, per_tile * 2400 / slabs, mf))
    del x, w, y
print("[%s] " % label + " | ".join(row))
