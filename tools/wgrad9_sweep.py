"""Nine-tap weight gradient of the branch groups (B = 8) under hrseg_tune wgrad9_blocks (target blocks per problem; 0 = the
library's plan): isolated time of the 4-, 3- and 2-branch group calls (kernel + ordered reduce)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hrseg_amd import _lib, ops

dev = torch.device("cuda:0")
B = 8
sizes, chans = [155, 78, 39, 20], [48, 96, 192, 384]
pr = _lib.CONV_PRECISION["auto"]
xs = [torch.randn(B, h, h, c, device=dev) for c, h in zip(chans, sizes)]
dys = [torch.randn(B, h, h, c, device=dev) * 1e-3 for c, h in zip(chans, sizes)]
gms = [d.abs().max().reshape(1).repeat(64) for d in dys]
dws = [torch.zeros(c, 9, c, device=dev) for c in chans]


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


for rep in range(2):
    for blocks in [0] + [int(v) for v in sys.argv[1:]]:
        _lib.tune(wgrad9_blocks=blocks)
        row = []
        for n in (4, 3, 2, 1):
            row.append("%6.1f" % timeit(lambda: ops.conv_wgrad_group(xs[:n], dys[:n], dws[:n], 3, 1, prec=pr, gmaxs=gms[:n])))
        print("wgrad9_blocks=%4d   4b / 3b / 2b / 1b: %s us" % (blocks, " / ".join(row)))
_lib.tune(wgrad9_blocks=0)
