"""Wave-specialised branch convolutions (4-, 3- and 2-branch groups of the headline step, B=8): forward, data gradient and
ACCUMULATING data gradient, timed in isolation under hrseg_tune switches given as arguments ("key=value,key=value" each):
    python tools/ws_epilogue_ab.py "ws_epi_early=0" "ws_epi_early=1" "ws_epi_early=1,ws_epi_acc_cost=8" """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hrseg_amd import _lib, ops

dev = torch.device("cuda:0")
B = 8
sizes, chans = [155, 78, 39, 20], [48, 96, 192, 384]
pr = _lib.CONV_PRECISION["auto"]
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randn(B, h, h, c, device=dev, generator=g) for c, h in zip(chans, sizes)]
ws = [torch.randn(c, 9, c, device=dev, generator=g) * 0.05 for c in chans]
wts = [ops.weight_transpose(w, c, 9, c) for w, c in zip(ws, chans)]
dys = [torch.randn(B, h, h, c, device=dev, generator=g) * 1e-3 for c, h in zip(chans, sizes)]
gms = [d.abs().max().reshape(1).repeat(64) for d in dys]
outs = [torch.zeros_like(x) for x in xs]


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


configs = sys.argv[1:] or [""]
print("%-44s %s" % ("tune", "  ".join("%db fwd / dgrad / dgrad+acc" % n for n in (4, 3, 2))))
for rep in range(2):
    for cfg in configs:
        kv = {k: int(v) for k, v in (p.split("=") for p in cfg.split(",") if p)}
        _lib.tune(**kv)
        row = []
        for n in (4, 3, 2):
            shp = [tuple(x.shape) for x in xs[:n]]
            t_f = timeit(lambda: ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n], prec=pr))
            t_d = timeit(lambda: ops.conv_dgrad_group(dys[:n], wts[:n], shp, 3, 1, list(outs[:n]), [False] * n, prec=pr, gmaxs=gms[:n]))
            t_a = timeit(lambda: ops.conv_dgrad_group(dys[:n], wts[:n], shp, 3, 1, list(outs[:n]), [True] * n, prec=pr, gmaxs=gms[:n]))
            row.append("%6.1f / %6.1f / %6.1f" % (t_f, t_d, t_a))
        print("%-44s %s" % (cfg or "(defaults)", "   ".join(row)))
        _lib.tune(**{k: 0 for k in kv if k != "ws_epi_early"}, **({"ws_epi_early": 1} if "ws_epi_early" in kv else {}))
