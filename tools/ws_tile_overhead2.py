"""Per-tile and per-launch overhead of the wave-specialised kernel: one problem, Cin -> Cout at H x H, batch B varied so that
the tiles per persistent block vary: T = t_launch + tiles_per_block * t_tile(nks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hrseg_amd import _lib, ops

dev = torch.device("cuda:0")
pr = _lib.CONV_PRECISION["auto"]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, cin, cout in [(160, 48, 48), (160, 96, 48), (160, 96, 96), (160, 192, 96)]:
    rows = []
    for B in (32, 64):
        # 160 x 160: 20 x 10 tiles per image, no padding; B images -> 200 B tiles (x channel tiles): 25 / 50 per block
        tiles = B * 200 * (cout // (96 if cout % 96 == 0 else 48))
        x = torch.randn(B, H, H, cin, device=dev)
        w = torch.randn(cout, 9, cin, device=dev) * 0.05
        t = timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, prec=pr))
        rows.append((tiles // 256, t))
        del x
    n, t = np.array([r[0] for r in rows], float), np.array([r[1] for r in rows], float)
    slope, icpt = np.polyfit(n, t, 1)
    nks = cin // 48
    print(f"{cin}->{cout} @ {H}: " + ", ".join(f"{int(a)} tiles/block {b:.1f} us" for a, b in rows) +
          f"  => {slope:.2f} us per tile ({nks * 14} slabs), {icpt:.1f} us per launch")
