"""Per-tile overhead of the wave-specialised 3x3 kernel: single-problem launches at one image size with Cin = 48 * nks
(nks K stages of 14 slabs per tile) -> time per tile = nks * 14 * t_slab + t_tile; a linear fit over nks separates the two.
    python tools/ws_tile_overhead.py [H] [Cout]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hrseg_amd import _lib, ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 155
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 48
B = 8
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


tiles = B * ((H + 7) // 8) * ((H + 15) // 16) * (Cout // (96 if Cout % 96 == 0 else 48))
per_block = -(-tiles // 256)
rows = []
for nks in (1, 2, 3, 4, 6, 8):
    cin = 48 * nks
    x = torch.randn(B, H, H, cin, device=dev)
    w = torch.randn(Cout, 9, cin, device=dev) * 0.05
    _lib.launch_count(None, reset=True)
    t = timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, prec=pr))
    fam = {f: _lib.launch_count(f) for f in ("ws", "patch_sp", "sp_im2col")}
    rows.append((nks, t))
    print(f"Cin {cin:4d} -> {Cout}: {t:8.1f} us   ({tiles} tiles, {per_block} per block; {fam})")
n, t = np.array([r[0] for r in rows], float), np.array([r[1] for r in rows], float)
slope, icpt = np.polyfit(n, t, 1)
print(f"fit: {slope / per_block / 14 * 1e3:.1f} ns per slab, {icpt / per_block:.2f} us per tile beyond its slabs "
      f"(= {icpt / per_block / (slope / per_block / 14):.1f} slab times)")
