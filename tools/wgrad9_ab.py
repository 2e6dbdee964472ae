"""A/B of the nine-tap weight-gradient variants on the four-branch group (B=8): block-synchronous vs role-split body,
with and without the XCD-contiguous block order.  python tools/wgrad9_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrseg_amd import _lib, ops  # noqa: E402


def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda")
    chans, sizes, B = [48, 96, 192, 384], [155, 78, 39, 20], 8
    pr = _lib.CONV_PRECISION["fp16x2"]
    for n in (4, 3, 2):
        xs = [torch.randn(B, h, h, c, device=dev) for c, h in zip(chans[:n], sizes[:n])]
        dys = [torch.randn(B, h, h, c, device=dev) * 1e-4 for c, h in zip(chans[:n], sizes[:n])]
        dws = [torch.zeros(c, 9, c, device=dev) for c in chans[:n]]
        gms = [d.abs().max().reshape(1).repeat(64) for d in dys]
        flops = sum(2.0 * B * h * h * c * c * 9 for c, h in zip(chans[:n], sizes[:n]))
        for ws in (0, 1):
            for xcd in (0, 1):
                _lib.tune(wgrad9_ws=ws, wgrad9_xcd=xcd)
                t = timed(lambda: ops.conv_wgrad_group(xs, dys, dws, 3, 1, prec=pr, gmaxs=gms))
                print(f"branches {n}  role_split {ws}  xcd_order {xcd}: {t:7.1f} us  {flops / t / 1e6:6.1f} TFLOP/s", flush=True)
    _lib.tune(wgrad9_ws=0, wgrad9_xcd=0)


if __name__ == "__main__":
    main()
