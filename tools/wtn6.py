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
    return e0.elapsed_time(e1) / n * 1e3
for c, h in ((96, 78), (192, 39), (384, 20), (96, 155)):
    x = torch.randn(4, h, h, c, device="cuda"); w = torch.randn(c, 9, c, device="cuda") * 0.05
    y = ops.conv_fwd(x, w, None, 3, 1); ref = y.clone()
    fl = 2.0 * y.numel() * c * 9
    for tune in ((0, 0, 0, 0), (1, 3, 1, 1), (11, 3, 1, 1), (11, 1, 1, 1), (12, 1, 1, 1), (12, 3, 1, 1)):
        _lib.set_conv_tune(*tune)
        t = timeit(lambda: ops.conv_fwd(x, w, None, 3, 1, out=y))
        _lib.set_conv_tune(*tune)
        ops.conv_fwd(x, w, None, 3, 1, out=y)
        print("C=%d H=%d tune=%s %.1f us %.0f%%  maxdiff %.2g" % (c, h, tune, t, 100 * fl / t / 157.3e6, float((y - ref).abs().max())), flush=True)
    _lib.set_conv_tune()
