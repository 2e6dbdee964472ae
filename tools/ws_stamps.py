"""Who waits for whom at the slab barrier of the wave-specialised kernel.  Needs a MEASUREMENT build of conv_ws.hip
(-DHRSEG_WS_STAMP=1): block 0's consumer wave 0 and producer wave 4 write s_memtime when they arrive at each slab barrier and
when they leave it.     python tools/ws_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hrseg_amd import _lib, ops

import ctypes

ops._stat_buffer = lambda cout, device: (torch.zeros(ops.STAT_ROWS_MAX * 2 * cout, dtype=torch.float64, device=device), ctypes.c_int(0))   # (stamps land in zeros)
dev = torch.device("cuda:0")
pr = _lib.CONV_PRECISION["auto"]
g = torch.Generator(device="cuda").manual_seed(1)
for c, h, b, nsl in [(48, 155, 64, 14), (96, 78, 128, 14), (384, 78, 32, 14), (64, 78, 128, 18)]:
    x = torch.randn(b, h, h, c, device=dev, generator=g)
    w = torch.randn(c, 9, c, device=dev, generator=g) * 0.05
    for _ in range(2):
        y, st = ops.conv_fwd(x, w, None, 3, 1, prec=pr, stats=True)
    torch.cuda.synchronize()
    part = st[0]
    raw = part.view(torch.int64).reshape(-1)[: 2 * 1024 * 2].cpu().numpy().reshape(2, 1024, 2)
    n = 6 * nsl
    ca, cl = raw[0, 1:n, 0], raw[0, 1:n, 1]          # consumer arrive / leave (slab barriers; entry 0 is the first)
    pa, pl = raw[1, 1:n, 0], raw[1, 1:n, 1]
    t0 = cl[0]
    print("== %d @ %d: per slab [consumer busy, consumer wait | producer busy, producer wait] cycles" % (c, h))
    cb = ca[1:] - cl[:-1]
    cw = cl[1:] - ca[1:]
    pb = pa[1:] - pl[:-1]
    pw = pl[1:] - pa[1:]
    for j in range(0, min(len(cb), 5 * nsl)):
        tag = " <- tile/stage end" if (j + 2) % nsl == 0 else ""
        print("  slab %3d: C %5d %5d | P %5d %5d%s" % (j + 1, cb[j], cw[j], pb[j], pw[j], tag))
    ep = part.view(torch.int64).reshape(-1)[4096 * 1 : 4096 + 1024 * 4].cpu().numpy().reshape(1024, 4)
    ep = ep[(ep[:, 0] > 0) & (ep[:, 3] > ep[:, 0])][1:6]
    for e in ep:
        print("  epilogue of a tile: values to add %5d | scale, (statistics,) stores %5d | zero the accumulators %5d cycles" % (
            e[1] - e[0], e[2] - e[1], e[3] - e[2]))
    print("  mean over %d slabs: consumer busy %.0f wait %.0f | producer busy %.0f wait %.0f | slab %.0f" % (
        len(cb), cb.mean(), cw.mean(), pb.mean(), pw.mean(), (cl[-1] - cl[0]) / (len(cl) - 1)))
