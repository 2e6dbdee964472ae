"""A/B of the wide-tile im2col body (pre-split weights, igemm_spw_body) against the narrow one on the single-launch
1x1 / stride-2 layers of one HRNet-W48 step at 620x620, B=8.  python tools/spw_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrseg_amd import _lib, ops  # noqa: E402


def timed(fn, reps=20):
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
    pr = _lib.CONV_PRECISION["auto"]
    B = 8
    cases = [(720, 720, 1, 1, 155), (64, 256, 1, 1, 155), (256, 64, 1, 1, 155), (256, 96, 3, 2, 155), (256, 48, 3, 1, 155)]
    for cin, cout, k, s, h in cases:
        x = torch.randn(B, h, h, cin, device=dev)
        w = torch.randn(cout, k * k, cin, device=dev) * 0.05
        ho = (h + 2 * ((k - 1) // 2) - k) // s + 1
        dy = torch.randn(B, ho, ho, cout, device=dev) * 1e-3
        gm = dy.abs().max().reshape(1).repeat(64)
        wt = ops.weight_transpose(w, cout, k * k, cin)
        flops = 2.0 * B * ho * ho * cin * cout * k * k
        for wide in (0, 1):
            _lib.tune(sp_wide=wide)
            _lib.launch_count(None, reset=True)
            tf = timed(lambda: ops.conv_fwd(x, w, None, k, s, prec=pr))
            nf = _lib.launch_count("sp_wide", reset=True)
            tb = timed(lambda: ops.conv_dgrad(dy, wt, x.shape, k, s, prec=pr, gmax=gm))
            nb = _lib.launch_count("sp_wide", reset=True)
            print(f"{cin:4d}->{cout:4d} k{k} s{s} @{h}  wide {wide}: fwd {tf:7.1f} us {flops / tf / 1e6:6.1f} TF ({'wide' if nf else 'other'})   "
                  f"dgrad {tb:7.1f} us {flops / tb / 1e6:6.1f} TF ({'wide' if nb else 'other'})", flush=True)
    _lib.tune(sp_wide=1)


if __name__ == "__main__":
    main()
