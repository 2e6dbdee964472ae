"""Grouped vs separate launches of the four HRNet stage-4 branch convs (B=4, 620x620)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops
chans, sizes = [48, 96, 192, 384], [155, 78, 39, 20]
xs = [torch.randn(4, h, h, c, device="cuda") for c, h in zip(chans, sizes)]
ws = [torch.randn(c, 9, c, device="cuda") * 0.05 for c in chans]
flops = sum(2.0 * x.numel() * c * 9 for x, c in zip(xs, chans))
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
ys = ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans)
dws = [torch.zeros_like(w) for w in ws]
t_sep = timeit(lambda: [ops.conv_fwd(x, w, None, 3, 1, out=y) for x, w, y in zip(xs, ws, ys)])
t_grp = timeit(lambda: ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans))
tw_sep = timeit(lambda: [ops.conv_wgrad(x, y, dw, 3, 1) for x, y, dw in zip(xs, ys, dws)])
tw_grp = timeit(lambda: ops.conv_wgrad_group(xs, ys, dws, 3, 1))
for n in (2, 3):
    t_n = timeit(lambda: ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n]))
    f_n = sum(2.0 * x.numel() * c * 9 for x, c in zip(xs[:n], chans[:n]))
    print("fwd group of %d: %.1f us %.1f TF" % (n, t_n * 1e6, f_n / t_n / 1e12))
print("fwd  separate %.1f us (%.1f TF)  grouped %.1f us (%.1f TF)" % (t_sep * 1e6, flops / t_sep / 1e12, t_grp * 1e6, flops / t_grp / 1e12))
print("wgrad separate %.1f us (%.1f TF)  grouped %.1f us (%.1f TF)" % (tw_sep * 1e6, flops / tw_sep / 1e12, tw_grp * 1e6, flops / tw_grp / 1e12))
