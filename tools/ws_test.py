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
for c, h, k in ((48, 155, 3), (96, 78, 3), (192, 39, 3), (64, 155, 3), (64, 620, 3), (256, 155, 3), (720, 155, 1)):
    x = torch.randn(4, h, h, c, device="cuda"); w = torch.randn(c, k * k, c, device="cuda") * 0.05
    _lib.set_conv_tune()
    y = ops.conv_fwd(x, w, None, k, 1); ref = y.clone()
    fl = 2.0 * y.numel() * c * k * k
    out = []
    for tune in ((0, 0, 0, 0), (1, 1, 3, 1), (2, 1, 3, 1), (4, 1, 3, 1), (12, 1, 3, 1)):
        _lib.set_conv_tune(*tune)
        t = timeit(lambda: ops.conv_fwd(x, w, None, k, 1, out=y))
        ops.conv_fwd(x, w, None, k, 1, out=y)
        out.append("%s %.1fus %.0f%% (d=%.1g)" % (tune[0] if tune[0] else "auto", t, 100 * fl / t / 157.3e6, float((y - ref).abs().max())))
    _lib.set_conv_tune()
    print("C=%d H=%d k=%d | %s" % (c, h, k, " | ".join(out)), flush=True)
