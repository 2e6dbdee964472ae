"""Repeatability stress of the wave-specialised kernels (their results do not depend on timing: no atomics without statistics):
every case N times, each result compared bit for bit with the first.    python tools/ws_stress.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hrseg_amd import _lib, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
pr = _lib.CONV_PRECISION["auto"]
g = torch.Generator(device="cuda").manual_seed(3)
bad = 0
# single problems: (Cin, Cout, H, W, B)
for cin, cout, H, W, B in [(48, 48, 155, 155, 8), (96, 96, 78, 78, 8), (192, 192, 39, 39, 8), (384, 384, 20, 20, 8), (144, 96, 78, 61, 5),
                           (64, 64, 155, 155, 4), (128, 64, 70, 61, 6), (48, 96, 46, 92, 6), (96, 48, 37, 45, 13), (240, 48, 100, 90, 3)]:
    x = torch.randn(B, H, W, cin, device=dev, generator=g)
    w = torch.randn(cout, 9, cin, device=dev, generator=g) * 0.05
    wt = ops.weight_transpose(w, cout, 9, cin)
    dy = torch.randn(B, H, W, cout, device=dev, generator=g) * 1e-3
    gm = dy.abs().max().reshape(1).repeat(64)
    y0 = ops.conv_fwd(x, w, None, 3, 1, prec=pr).clone()
    d0 = ops.conv_dgrad(dy, wt, x.shape, 3, 1, prec=pr, gmax=gm).clone()
    nf = nd = 0
    for _ in range(N):
        nf += int(not torch.equal(ops.conv_fwd(x, w, None, 3, 1, prec=pr), y0))
        nd += int(not torch.equal(ops.conv_dgrad(dy, wt, x.shape, 3, 1, prec=pr, gmax=gm), d0))
    print("%4d -> %4d  %3dx%3d B=%2d : forward %d / %d differ, data gradient %d / %d differ, finite %s" % (
        cin, cout, H, W, B, nf, N, nd, N, bool(torch.isfinite(y0).all() and torch.isfinite(d0).all())), flush=True)
    bad += nf + nd
# the four-branch group of the headline step
sizes, chans = [155, 78, 39, 20], [48, 96, 192, 384]
xs = [torch.randn(8, h, h, c, device=dev, generator=g) for c, h in zip(chans, sizes)]
ws = [torch.randn(c, 9, c, device=dev, generator=g) * 0.05 for c in chans]
for n in (4, 3, 2):
    y0 = [y.clone() for y in ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n], prec=pr)]
    nf = 0
    for _ in range(N):
        ys = ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n], prec=pr)
        nf += int(not all(torch.equal(a, b) for a, b in zip(ys, y0)))
    print("group of %d: forward %d / %d differ" % (n, nf, N), flush=True)
    bad += nf
print("TOTAL differing results:", bad)
sys.exit(1 if bad else 0)
