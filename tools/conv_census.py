"""Which convolution calls does one train step make, and what does each cost?  Records every ops.conv_* call of one
step of the bench model (shapes only), then times each distinct call in isolation (same arithmetic mode).
    python tools/conv_census.py [--size 620] [--batch 4]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hrseg_amd import _lib, ops
from hrseg_amd import train as T
from hrseg_amd.utils import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=620)
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
args = argparse.Namespace(model="hrnet", size=a.size, batch=a.batch, flat=False, tree="class_tree_tl.json")
dev = torch.device("cuda:0")
tree, model, ns, loss_fns, opt = bench.build(args, dev)
x, t = synth.synthetic_batch(tree, a.batch, a.size, seed=1, hierarchical=True)
x, t = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
model.train()
T.train_step(model, opt, x, t, loss_fns, ns, tree, [])          # warm-up (allocations)
torch.cuda.synchronize()

calls = collections.Counter()
orig = {n: getattr(ops, n) for n in ("conv_fwd", "conv_dgrad", "conv_wgrad", "conv_fwd_group", "conv_dgrad_group", "conv_wgrad_group")}


def sig(t):
    return tuple(t.shape)


def w_fwd(x, w, bias, k, s, out=None, cout=None, prec=0):
    calls[("fwd", (sig(x),), (cout if cout is not None else w.shape[0],), k, s, prec)] += 1
    return orig["conv_fwd"](x, w, bias, k, s, out=out, cout=cout, prec=prec)


def w_dgrad(dy, wt, x_shape, k, s, out=None, accumulate=False, prec=0, gmax=None):
    calls[("dgrad", (tuple(x_shape),), (dy.shape[3],), k, s, prec)] += 1
    return orig["conv_dgrad"](dy, wt, x_shape, k, s, out=out, accumulate=accumulate, prec=prec, gmax=gmax)


def w_wgrad(x, dy, dw, k, s, prec=0, gmax=None):
    calls[("wgrad", (sig(x),), (dy.shape[3],), k, s, prec)] += 1
    return orig["conv_wgrad"](x, dy, dw, k, s, prec=prec, gmax=gmax)


def w_fwd_g(xs, ws, biases, k, s, couts, prec=0):
    calls[("fwd", tuple(sig(x) for x in xs), tuple(couts), k, s, prec)] += 1
    return orig["conv_fwd_group"](xs, ws, biases, k, s, couts, prec=prec)


def w_dgrad_g(dys, wts, x_shapes, k, s, outs, accumulate, prec=0, gmaxs=None):
    calls[("dgrad", tuple(tuple(xs) for xs in x_shapes), tuple(d.shape[3] for d in dys), k, s, prec)] += 1
    return orig["conv_dgrad_group"](dys, wts, x_shapes, k, s, outs, accumulate, prec=prec, gmaxs=gmaxs)


def w_wgrad_g(xs, dys, dws, k, s, prec=0, gmaxs=None):
    if len(xs) > 1 or not (prec and k == 3 and s == 1):
        calls[("wgrad", tuple(sig(x) for x in xs), tuple(d.shape[3] for d in dys), k, s, prec)] += 1
    return orig["conv_wgrad_group"](xs, dys, dws, k, s, prec=prec, gmaxs=gmaxs)


ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = w_fwd, w_dgrad, w_wgrad
ops.conv_fwd_group, ops.conv_dgrad_group, ops.conv_wgrad_group = w_fwd_g, w_dgrad_g, w_wgrad_g
import hrseg_amd.engine as E
T.train_step(model, opt, x, t, loss_fns, ns, tree, [])
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(ops, n, f)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
for (kind, xshapes, couts, k, s, prec), cnt in calls.items():
    xs = [torch.randn(sh, device=dev) for sh in xshapes]
    ws = [torch.randn(co, k * k, sh[3], device=dev) * 0.05 for co, sh in zip(couts, xshapes)]
    ys = [ops.conv_fwd(x_, w_, None, k, s) for x_, w_ in zip(xs, ws)]
    fl = sum(2.0 * y.numel() * sh[3] * k * k for y, sh in zip(ys, xshapes))
    n = len(xs)
    dys = [torch.randn_like(y) for y in ys]
    gms = [d.abs().max().reshape(1).repeat(64) for d in dys]
    if kind == "fwd":
        fn = (lambda: ops.conv_fwd(xs[0], ws[0], None, k, s, prec=prec)) if n == 1 else (lambda: ops.conv_fwd_group(xs, ws, [None] * n, k, s, list(couts), prec=prec))
    elif kind == "dgrad":
        wts = [ops.weight_transpose(w_, co, k * k, sh[3]) for w_, co, sh in zip(ws, couts, xshapes)]
        fn = (lambda: ops.conv_dgrad(dys[0], wts[0], xshapes[0], k, s, prec=prec, gmax=gms[0])) if n == 1 else (
            lambda: ops.conv_dgrad_group(dys, wts, list(xshapes), k, s, [None] * n, [False] * n, prec=prec, gmaxs=gms))
    else:
        dws = [torch.zeros_like(w_) for w_ in ws]
        fn = (lambda: ops.conv_wgrad(xs[0], dys[0], dws[0], k, s, prec=prec, gmax=gms[0])) if n == 1 else (
            lambda: ops.conv_wgrad_group(xs, dys, dws, k, s, prec=prec, gmaxs=gms))
    try:
        t_us = timeit(fn)
    except Exception as e:          # a shape the single-call path does not take
        t_us = float("nan")
    rows.append((cnt * t_us, kind, cnt, t_us, fl, xshapes, couts, k, s))
    del xs, ws, ys, dys
rows.sort(reverse=True)
tot = sum(r[0] for r in rows if r[0] == r[0])
print("total %.2f ms of conv calls per step" % (tot / 1e3))
for tt, kind, cnt, t_us, fl, xshapes, couts, k, s in rows[:45]:
    desc = " + ".join("%dx%dx%d->%d" % (sh[1], sh[2], sh[3], co) for sh, co in zip(xshapes, couts))
    print("%6.2f ms  %-5s x%3d  %8.1f us  %6.1f TF  k%d s%d B%d  %s" % (tt / 1e3, kind, cnt, t_us, fl / t_us / 1e6, k, s, xshapes[0][0], desc))
