"""Grid of the grouped weight gradient (blocks per problem = clamp(mult * tiles, min, max)) at the executed batch."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops, _lib
B = int(os.environ.get("SWEEP_B", "8"))
chans, sizes = [48, 96, 192, 384], [155, 78, 39, 20]
xs = [torch.randn(B, h, h, c, device="cuda") for c, h in zip(chans, sizes)]
dys = [torch.randn(B, h, h, c, device="cuda") for c, h in zip(chans, sizes)]
dws = [torch.zeros(c, 9, c, device="cuda") for c in chans]
fl = [2.0 * B * h * h * c * c * 9 for c, h in zip(chans, sizes)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(n):
    f = lambda: ops.conv_wgrad_group(xs[:n], dys[:n], dws[:n], 3, 1)
    f(); f()
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e-3
for mult, mn, mx in [(7, 512, 4096), (3, 768, 2048), (2, 768, 4096), (2, 768, 2048), (3, 768, 3072), (4, 768, 2048), (3, 640, 2048), (3, 896, 2048), (2, 768, 1536), (3, 768, 1728), (2, 896, 2304)]:
    _lib.set_wgrad_group_plan(mult, mn, mx)
    r = [t(n) for n in (2, 3, 4)]
    print("mult %d min %4d max %4d: " % (mult, mn, mx) + "  ".join("n=%d %6.1f us %5.1f TF" % (n, x * 1e6, sum(fl[:n]) / x / 1e12) for n, x in zip((2, 3, 4), r)))
