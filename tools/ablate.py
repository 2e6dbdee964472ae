"""Timing ablation of igemm: normal / no global loads after stage 0 / no MFMA."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops, _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for c, h, B in ((48, 155, 4), (96, 78, 4), (64, 155, 4), (64, 620, 4)):
    x = torch.randn(B, h, h, c, device="cuda"); w = torch.randn(c, 9, c, device="cuda") * 0.05
    y = ops.conv_fwd(x, w, None, 3, 1)
    fl = 2.0 * y.numel() * c * 9
    for tune in ((0, 0, 0, 0), (1, 3, 1, 1), (2, 1, 1, 1)):
        if c % 48 and tune[1] == 3: continue
        _lib.set_conv_tune(*tune)
        r = []
        for mode in (0, 1, 2, 3, 4):
            lib.hrseg_debug_set_conv_ablation(mode)
            r.append(timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, out=y)))
        lib.hrseg_debug_set_conv_ablation(0)
        print("C=%d H=%d tune=%s ideal %.1fus | normal %.1f | noload %.1f | nomfma %.1f | +nobarrier/nowrite %.1f | +noldsread %.1f" % (c, h, tune, fl / 157.3e12 * 1e6, *r), flush=True)
    _lib.set_conv_tune()
