"""Does a second stream pay in the forward?  One stage-4 HighResolutionModule's branch section (4 BasicBlocks per branch:
conv-BN-ReLU-conv-BN-add-ReLU) at the headline geometry (B=8: two batched level passes of 4 images), issued
  (a) as the engine does: every layer ONE grouped launch for the four branches, one stream
  (b) branch 0 (155x155x48: 3/4 of the BatchNorm bytes, 1/4 of the FLOPs) on one stream, branches 1-3 on a second one
python tools/branch_overlap.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrseg_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
chans, sizes, B = [48, 96, 192, 384], [155, 78, 39, 20], 8
pr = _lib.CONV_PRECISION["auto"]
xs0 = [torch.randn(B, h, h, c, device=dev) for c, h in zip(chans, sizes)]
ws = [[(torch.randn(c, 9, c, device=dev) * 0.05) for c in chans] for _ in range(8)]


def bn_items(ys, res, idx):
    return [dict(y=y, gamma=torch.ones(chans[i], device=dev), beta=torch.zeros(chans[i], device=dev),
                 rm=torch.zeros(chans[i], device=dev), rv=torch.ones(chans[i], device=dev),
                 nbt=torch.zeros((), dtype=torch.int64, device=dev), momentum=0.1, eps=1e-5,
                 residual=None if res is None else res[j], relu=True) for j, (y, i) in enumerate(zip(ys, idx))]


def chain(idx):
    xs = [xs0[i] for i in idx]
    n = len(idx)
    for blk in range(4):
        w1, w2 = [ws[2 * blk][i] for i in idx], [ws[2 * blk + 1][i] for i in idx]
        cs = [chans[i] for i in idx]
        ys = ops.conv_fwd_group(xs, w1, [None] * n, 3, 1, cs, prec=pr) if n > 1 else [ops.conv_fwd(xs[0], w1[0], None, 3, 1, prec=pr)]
        zs = [z for z, _ in ops.bn_fwd_group(bn_items(ys, None, idx), True)]
        ys = ops.conv_fwd_group(zs, w2, [None] * n, 3, 1, cs, prec=pr) if n > 1 else [ops.conv_fwd(zs[0], w2[0], None, 3, 1, prec=pr)]
        xs = [z for z, _ in ops.bn_fwd_group(bn_items(ys, xs, idx), True)]
    return xs


side = torch.cuda.Stream()


def one_stream():
    chain([0, 1, 2, 3])


def two_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain([1, 2, 3])
    chain([0])
    main.wait_stream(side)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in (("one stream, grouped", one_stream), ("two streams: {0} | {1,2,3}", two_streams), ("one stream, grouped", one_stream),
                 ("two streams: {0} | {1,2,3}", two_streams)):
    print(f"{name:32s} {timed(fn):8.1f} us per module branch section (forward)", flush=True)
