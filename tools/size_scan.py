"""Efficiency of one conv shape vs problem size (batch) -- ramp/tail or steady state?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops, _lib
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for c, h in ((48, 155), (96, 78), (384, 20)):
    w = torch.randn(c, 9, c, device="cuda") * 0.05
    for B in (1, 2, 4, 8, 16, 32, 64):
        x = torch.randn(B, h, h, c, device="cuda")
        y = ops.conv_fwd(x, w, None, 3, 1)
        fl = 2.0 * y.numel() * c * 9
        res = []
        for tune in ((0, 0, 0, 0), (1, 3, 1, 1), (2, 1, 1, 1), (2, 3, 1, 1), (4, 1, 1, 1)):
            _lib.set_conv_tune(*tune)
            t = timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, out=y))
            res.append("%s %.0fus %.0f%%" % ("auto" if tune[0] == 0 else "w%dk%d" % tune[:2], t * 1e6, 100 * fl / t / 157.3e12))
        _lib.set_conv_tune()
        print("C=%d H=%d B=%2d blocks64=%6d | %s" % (c, h, B, (B * h * h + 63) // 64 * (c // 48), " | ".join(res)), flush=True)
