"""Execution engine of the hot path: flat parameter storage, a forward recorder and
a hand-rolled reverse pass that launches the HIP kernels (ops.py).

Why not torch.autograd per op: a hierarchical HRNet step is ~5,000 kernel
launches; the engine keeps the bookkeeping to a Python list of closures, lets
weight gradients accumulate straight into one flat fp32 gradient buffer (the
L backbone passes add into the same slots, the RCCL all-reduce and the fused
AdamW run over that one buffer) and frees activations as the reverse pass
consumes them.  torch.autograd sees the whole model as ONE node
(Models/models.py:_Bridge), so ``loss.backward()`` of the reference loop works.

Activations are NHWC fp32 (`Act.data`), possibly a channel slice of a wider
tensor.  Gradient buffers are exclusively owned by one Act (no aliasing), so
in-place accumulation is always safe.
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import _lib, ops

# Weight gradients are off the critical path of the reverse pass (nothing reads them before the
# all-reduce / optimizer), so they run on a second HIP stream next to the data-gradient + BN chain:
# their MFMA work fills the time the chain spends in bandwidth-bound kernels.  HRSEG_WGRAD_STREAM=0
# keeps everything on one stream.
_SIDE = {}


_SIDE_ENABLED = None
_RELU_MASK = os.environ.get("HRSEG_BN_RELU_MASK", "1") != "0"      # 0: the backward of residual layers reads z for its ReLU mask
# MEASUREMENT ONLY (results are WRONG on purpose; tools/bn_fusion_bound.sh): upper bounds of what a conv <-> BatchNorm fusion
# could save, measured by leaving the launches out of the real step instead of estimating them on paper --
#   skip_apply1: no BN-apply launch for tensors with a single convolution reader (conv1 -> conv2 of a BasicBlock): the reader
#                takes the raw conv output (same bytes, same kernels) = the ceiling of "BN-apply + ReLU in the consumer's staging"
#   skip_stats:  no BN-statistics launch in the forward = the ceiling of "statistics in the convolution epilogue"
#   skip_wgrad:  no weight-gradient launch at all = what the side stream's work costs the step beside the main chain
_EXPERIMENT = set(filter(None, os.environ.get("HRSEG_EXPERIMENT", "").split(",")))
_CONV_STATS = os.environ.get("HRSEG_CONV_STATS", "1") != "0"       # 0: BatchNorm statistics always as their own launch
# pre-split activations (include/hrseg.h x_split / z_split): 0 = fp32 everywhere; 1 = conv1 -> conv2 inside a block only (the
# tensor has no other reader: results are bit-identical); 2 (default) = also a block's output where the next block of the branch
# is its only reader -- that block's BatchNorm adds it as the residual in its 22-bit form (hi + lo), a 2^-23 relative rounding
_X_SPLIT = int(os.environ.get("HRSEG_X_SPLIT", "2"))


def wgrad_stream(device):
    global _SIDE_ENABLED
    if _SIDE_ENABLED is None:            # read once: this runs per conv group
        _SIDE_ENABLED = os.environ.get("HRSEG_WGRAD_STREAM", "1") != "0"
    if not _SIDE_ENABLED:
        return None
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


class Act:
    """An activation and its gradient slot.  `split`: the tensor is stored pre-split for fp16x2 convolutions (per 4 channels
    {hi01, hi23, lo01, lo23}, include/hrseg.h x_split) -- only convolutions read it, through conv_bn_group"""
    __slots__ = ("data", "grad", "needs_grad", "slot", "split")

    def __init__(self, data, needs_grad=True, split=False):
        self.data = data
        self.grad = None
        self.needs_grad = needs_grad
        self.split = split

    @property
    def shape(self):
        return self.data.shape


class FlatParams:
    """All parameters of a module as views of one flat fp32 buffer (+ a flat gradient).

    Conv weights keep their logical [Cout,Cin,kh,kw] shape but live in OHWI order
    (channels_last strides), which is what the conv kernels read; state_dict()
    and load_state_dict() keep working on the logical shape.
    """

    def __init__(self, module, device):
        self.device = device
        params = [(n, p) for n, p in module.named_parameters()]
        sizes = [p.numel() for _, p in params]
        # every slot starts on a 16-byte boundary (vector loads of weights/bias)
        offs, total = [], 0
        for s in sizes:
            offs.append(total)
            total += (s + 3) // 4 * 4
        self.numel = total
        self.data = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.slots = {}
        self.params = []
        with torch.no_grad():
            for (name, p), off, n in zip(params, offs, sizes):
                store = self.data[off:off + n]
                gstore = self.grad[off:off + n]
                if p.dim() == 4:
                    o, i, kh, kw = p.shape
                    store.view(o, kh, kw, i).copy_(p.detach().permute(0, 2, 3, 1).to(device))
                    view = store.view(o, kh, kw, i).permute(0, 3, 1, 2)
                    gview = gstore.view(o, kh, kw, i).permute(0, 3, 1, 2)
                else:
                    store.view(p.shape).copy_(p.detach().to(device))
                    view, gview = store.view(p.shape), gstore.view(p.shape)
                p.data = view
                p._hr_store, p._hr_gstore, p._hr_gview = store, gstore, gview
                p._hr_range = (off, off + n)
                self.slots[name] = (off, n)
                self.params.append(p)
        self._first, self._last = self.params[0], self.params[-1]
        self.grads_fresh = True   # flat grad is all zeros
        # transposed copies of every conv weight (the data-gradient operand), refreshed once per step
        import numpy as np
        entries = []
        for (name, p), off in zip(params, offs):
            if p.dim() != 4:
                continue
            o, i, kh, kw = p.shape
            if i % 16 or o % 16:
                continue                                   # image layer / heads: no MFMA data-gradient
            for t in range(kh * kw):
                for co0 in range(0, o, 32):
                    for ci0 in range(0, i, 32):
                        entries.append((off, o, kh * kw, i, co0, ci0, t, 0))
            p._hr_tstore_range = (off, off + p.numel())
        self.data_t = torch.empty_like(self.data)
        self._wt_table = torch.from_numpy(np.asarray(entries, dtype=np.int32).reshape(-1)).to(device) if entries else None
        self._wt_entries = len(entries)
        self.wt_stale = True

    def transpose_all(self):
        if self.wt_stale:
            if self._wt_table is not None:
                ops.call("hrseg_weight_transpose_all", ops.ptr(self.data), ops.ptr(self.data_t), ops.ptr(self._wt_table),
                         self._wt_entries)
            self.wt_stale = False

    def transposed(self, p):
        """[Cin][taps][Cout] copy of conv weight p (valid until the next forward)"""
        self.transpose_all()
        lo, hi = p._hr_tstore_range
        return self.data_t[lo:hi]

    def refresh_weight_images(self, backward):
        """start of a model call: the pre-split weight images the wave-specialised kernels read are rebuilt for the current
        parameters by ONE launch (hrseg_weight_images_refresh) instead of one small launch in front of every convolution;
        `backward`: the call will run a reverse pass, so the transposed weights (the data gradients' operand) are produced
        first and their images refreshed with the rest.  -> whether the convolutions may flag their weights as persistent"""
        if not _lib.ensure_image_arena(self):
            return False
        if backward:
            self.transpose_all()
        ops.call("hrseg_weight_images_refresh")
        return True

    def view_of(self, flat_tensor, p):
        """the slice of another flat buffer (optimizer moments ...) that belongs to parameter p, in p's
        logical shape (conv weights: [Cout,Cin,kh,kw] view of the OHWI storage)"""
        lo, hi = p._hr_range
        store = flat_tensor[lo:hi]
        if p.dim() == 4:
            o, i, kh, kw = p.shape
            return store.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return store.view(p.shape)

    def valid(self, device):
        """False once .to()/.cuda() rebound the parameters to fresh storages."""
        return (self.device == device and self._first.data_ptr() == self.data.data_ptr()
                and self._last._hr_store.data_ptr() == self._last.data_ptr())

    def attach_grads(self):
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != p._hr_gview.data_ptr():
                p.grad = p._hr_gview

    def prepare_backward(self):
        """zero_grad(set_to_none=True) detaches the views: the flat gradient then restarts from zero."""
        if self._first.grad is None and not self.grads_fresh:
            ops.fill(self.grad, 0.0)
        self.grads_fresh = False


class Recorder:
    """Forward launcher + tape of backward closures for ONE model forward."""

    def __init__(self, training, record, flat=None, bn_repeat=1, bn_segments=1, prec=0, sync=None, fold=None, wpersist=False):
        self.wpersist = bool(wpersist) and flat is not None      # weights of the flat buffers have cached images (hrseg.h w_persistent)
        self.fold = fold                 # inference: (conv, bn) -> (folded weight, folded bias), or None = BN as its own launches
        self.sync = sync                 # process group for cross-rank BatchNorm statistics (opt-in sync_bn), else None
        self.prec = prec                 # _lib.CONV_PRECISION code of every convolution of this forward (and its backward)
        self.training = training
        self.record = record
        self.bn_repeat = bn_repeat       # running-stat updates per BN (level passes run as one)
        self.bn_segments = bn_segments   # the batch holds this many level passes of the same images (batched passes)
        self.tape = []
        self.flat = flat
        self.wt_cache = {}
        self.used_side = False

    # ------------------------------------------------------------------ helpers
    def _push(self, fn):
        if self.record:
            self.tape.append(fn)

    def mark(self, label):
        """a named position in the tape (DDP bucket boundaries)"""
        self._push(label)

    def backward(self, hook=None):
        """run the tape in reverse; str entries are marks handed to `hook`."""
        while self.tape:
            fn = self.tape.pop()
            if isinstance(fn, str):
                if hook is not None:
                    _lib.host_call(lambda mark=fn: hook(mark))      # (a launch tape re-issues it at this position)
            else:
                fn()

    def _wt(self, conv):
        if conv.in_channels <= 8:
            return conv.weight._hr_store      # first layer: the direct data-gradient kernel reads the forward layout
        if self.flat is not None and hasattr(conv.weight, "_hr_tstore_range"):
            return self.flat.transposed(conv.weight)
        key = id(conv)
        wt = self.wt_cache.get(key)
        if wt is None:
            co, ci, kh, kw = conv.weight.shape
            wt = ops.weight_transpose(conv.weight._hr_store, co, kh * kw, ci)
            self.wt_cache[key] = wt
        return wt

    @staticmethod
    def _give(act, g):
        """hand a freshly written gradient tensor to `act` (first contribution)"""
        act.grad = g

    # ------------------------------------------------------------------ conv + BN (+residual) (+ReLU)
    def conv_bn(self, x, conv, bn, relu, residual=None, out=None, split_for=None):
        return self.conv_bn_group([(x, conv, bn, residual)], relu, outs=[out],
                                  split_for=[split_for] if split_for is not None else None)[0]

    def conv_bn_group(self, items, relu, outs=None, single_reader=False, split_for=None, split_level=1):
        """... split_for: one convolution per item that is the ONLY reader of that item's output (conv1 -> conv2 of a block).
        Where the library will run those readers on the kernels that take a pre-split pixel operand (hrseg_conv_x_split_ok:
        wave-specialised forward + nine-tap weight gradient, fp16x2 arithmetic) the BatchNorm apply writes the output
        pre-split -- the same bytes the readers' staging waves would compute from the fp32 tensor, computed once."""
        """items: list of (x, conv, bn, residual-or-None), independent of each other (the parallel
        HRNet branches, the fuse paths of a module; a single layer is a group of one); relu: one flag or
        one per item.  Per group: one conv launch, three BN launches; backward: three BN launches, one
        wgrad launch and one dgrad launch per round of problems with DISTINCT inputs (two problems
        that read the same tensor must not write its gradient in one launch)."""
        n = len(items)
        relus = list(relu) if isinstance(relu, (list, tuple)) else [relu] * n
        k, s = items[0][1].kernel_size[0], items[0][1].stride[0]
        assert all(c.kernel_size[0] == k and c.stride[0] == s for _, c, _, _ in items)
        xs = [it[0] for it in items]
        if not self.training and not self.record and self.fold is not None:
            # inference: BatchNorm folded into the weights (cached per conv until a parameter or a running statistic
            # changes), residual add and ReLU in the convolution's epilogue: ONE launch per layer group
            folded = [self.fold(conv, bn) for _, conv, bn, _ in items]
            res = [it[3].data if it[3] is not None else None for it in items]
            prec = self.prec if all(f[2] for f in folded) else _lib.CONV_PRECISION["f32"]    # (a folded weight out of fp16x2's range)
            if n == 1:
                conv = items[0][1]
                zs = [ops.conv_fwd(xs[0].data, folded[0][0], folded[0][1], k, s, cout=conv.out_channels, prec=prec,
                                   residual=res[0], relu=relus[0], out=outs[0] if outs is not None else None)]
            else:
                assert outs is None or all(o is None for o in outs)
                zs = ops.conv_fwd_group([x.data for x in xs], [f[0] for f in folded], [f[1] for f in folded], k, s,
                                        [c.out_channels for _, c, _, _ in items], prec=prec, residuals=res, relus=relus)
            return [Act(z) for z in zs]
        # training: the convolution kernels that can leave the BatchNorm partial sums of their output behind (the
        # wave-specialised 3x3 kernels: hrseg_conv_shape_t.stat_partial) -- the statistics launch is then skipped below
        want_stats = self.training and _CONV_STATS and self.sync is None and not _lib.deterministic() and not _EXPERIMENT
        stats = None
        if n == 1:
            conv = items[0][1]
            ys = ops.conv_fwd(xs[0].data, conv.weight._hr_store, conv.bias._hr_store if conv.bias is not None else None,
                              k, s, cout=conv.out_channels, prec=self.prec, stats=want_stats, wpersist=self.wpersist,
                              x_split=xs[0].split)
            if want_stats:
                ys, st = ys
                stats = [st]
            ys = [ys]
        else:
            ys = ops.conv_fwd_group([x.data for x in xs], [c.weight._hr_store for _, c, _, _ in items],
                                    [c.bias._hr_store if c.bias is not None else None for _, c, _, _ in items], k, s,
                                    [c.out_channels for _, c, _, _ in items], prec=self.prec, stats=want_stats,
                                    wpersist=self.wpersist, x_splits=[x.split for x in xs])
            if want_stats:
                ys, stats = ys
        if stats is not None and any(st is None for st in stats):
            stats = None                                 # (one statistics plan per grouped BatchNorm call)
        # layers with a residual and a ReLU: the backward's mask is not recomputable from y; the forward leaves it as one
        # byte per channel quad (hrseg_bn_fwd_t.relu_mask) so that the backward reads 1/16 of what reading z costs
        masks = [torch.empty((y.shape[0] * y.shape[1] * y.shape[2], y.shape[3] // 4), dtype=torch.uint8, device=y.device)
                 if (self.record and _RELU_MASK and relus[i] and it[3] is not None) else None
                 for i, (it, y) in enumerate(zip(items, ys))]
        z_split = False
        if split_for is not None and _X_SPLIT >= split_level and not _EXPERIMENT and (outs is None or all(o is None for o in outs)) \
                and all(c is not None for c in split_for):
            k2, s2 = split_for[0].kernel_size[0], split_for[0].stride[0]
            if all(c.kernel_size[0] == k2 and c.stride[0] == s2 for c in split_for):
                z_split = ops.conv_x_split_ok([tuple(y.shape) for y in ys], [c.out_channels for c in split_for], k2, s2, self.prec)
                # (a layer with residual + ReLU hands its backward the mask bytes; without them the backward would read z as fp32)
                z_split = z_split and all(m is not None or not (relus[i] and it[3] is not None)
                                          for i, (it, m) in enumerate(zip(items, masks)))
        bn_items = [dict(y=y, gamma=bn.weight._hr_store, beta=bn.bias._hr_store, rm=bn.running_mean, rv=bn.running_var,
                         nbt=bn.num_batches_tracked, momentum=bn.momentum, eps=bn.eps,
                         residual=res.data if res is not None else None, residual_split=bool(res is not None and res.split),
                         relu=relus[i], repeat=self.bn_repeat,
                         stat_div=self.bn_segments, relu_mask=masks[i],
                         out=outs[i] if outs is not None else None, z_split=z_split,
                         partial=stats[i] if stats is not None else None)
                    for i, ((x, conv, bn, res), y) in enumerate(zip(items, ys))]
        if _EXPERIMENT and self.training and self.sync is None:
            phases = 7
            if "skip_apply1" in _EXPERIMENT and single_reader:
                phases &= ~4
            if "skip_stats" in _EXPERIMENT:
                phases &= ~1
            zc = ops.bn_fwd_group(bn_items, self.training, phases=phases)
            if not phases & 4:
                zc = [(y, c) for y, (_, c) in zip(ys, zc)]
        else:
            zc = ops.bn_fwd_group(bn_items, self.training, sync=self.sync, phases=6 if stats is not None else 7)
        zs = [Act(z, split=z_split) for z, _ in zc]
        if not self.record:
            return zs
        coefs = [c for _, c in zc]
        eval_mode = not self.training

        want_gmax = self.prec in (_lib.CONV_PRECISION["fp16x2"], _lib.CONV_PRECISION["auto"])

        def bwd():
            bw = []
            # fp16x2: the BN backward records max|dy| per problem; the data / weight gradients scale dy by it
            gmax_all = torch.empty((n, 64), dtype=torch.float32, device=ys[0].device) if want_gmax else None    # reset by bn_bwd
            gmaxs = [gmax_all[i] for i in range(n)] if want_gmax else [None] * n
            for i, ((x, conv, bn, res), y, z, coef) in enumerate(zip(items, ys, zs, coefs)):
                dz = z.grad
                z.grad = None
                dres, dres_acc = None, False
                if res is not None and res.needs_grad:
                    if res.grad is None:
                        res.grad = torch.empty(res.data.shape, dtype=torch.float32, device=dz.device)
                    else:
                        dres_acc = True
                    dres = res.grad
                # without a residual the ReLU mask is recomputed from y: the backward never reads z
                bw.append(dict(dz=dz, z=z.data if (relus[i] and res is not None and masks[i] is None) else None,
                               relu_mask=masks[i], relu=relus[i], y=y, coef=coef,
                               dgamma=bn.weight._hr_gstore,
                               dbeta=bn.bias._hr_gstore, dres=dres, dres_accumulate=dres_acc, nseg=self.bn_segments,
                               dy_absmax=gmaxs[i]))
            dys = ops.bn_bwd_group(bw, eval_mode, sync=self.sync)
            # the weight gradient is issued on the side stream BEFORE the data gradient of the same layer (issuing it behind,
            # so that it would run beside the next BatchNorm backward, measured 55.9 vs 53.8 ms per step)
            def weight_gradients():
                if "skip_wgrad" in _EXPERIMENT:      # (measurement only: what the weight gradients cost the step beside the main chain)
                    return
                side = wgrad_stream(dys[0].device)
                if side is not None:
                    _lib.stream_wait(side, None)
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    if n == 1:
                        ops.conv_wgrad(xs[0].data, dys[0], items[0][1].weight._hr_gstore, k, s, prec=self.prec, gmax=gmaxs[0],
                                       x_split=xs[0].split)
                    else:
                        ops.conv_wgrad_group([x.data for x in xs], dys, [c.weight._hr_gstore for _, c, _, _ in items], k, s,
                                             prec=self.prec, gmaxs=gmaxs, x_splits=[x.split for x in xs])
                if side is not None:
                    if _lib.taping():
                        # a recorded step is replayed without the allocator: what the side stream reads stays allocated
                        # until the streams join at the end of the reverse pass (_Run.backward)
                        _lib.tape_keep(gmax_all, *dys, *[x.data for x in xs])
                    else:
                        if gmax_all is not None:
                            gmax_all.record_stream(side)
                        for t in dys:
                            t.record_stream(side)     # not reused before the side stream is done reading it
                        for x in xs:
                            x.data.record_stream(side)
                    self.used_side = True
            weight_gradients()
            # data gradients: rounds of problems whose inputs are distinct tensors
            todo = [i for i, x in enumerate(xs) if x.needs_grad]
            while todo:
                seen, rnd, rest = set(), [], []
                for i in todo:
                    (rest if id(xs[i]) in seen else rnd).append(i)
                    seen.add(id(xs[i]))
                if len(rnd) == 1:
                    i = rnd[0]
                    x = xs[i]
                    wp = self.wpersist and hasattr(items[i][1].weight, "_hr_tstore_range")
                    if x.grad is None:
                        x.grad = ops.conv_dgrad(dys[i], self._wt(items[i][1]), x.data.shape, k, s, prec=self.prec,
                                                gmax=gmaxs[i], wpersist=wp)
                    else:
                        ops.conv_dgrad(dys[i], self._wt(items[i][1]), x.data.shape, k, s, out=x.grad, accumulate=True,
                                       prec=self.prec, gmax=gmaxs[i], wpersist=wp)
                else:
                    got = ops.conv_dgrad_group([dys[i] for i in rnd], [self._wt(items[i][1]) for i in rnd],
                                               [xs[i].data.shape for i in rnd], k, s, [xs[i].grad for i in rnd],
                                               [xs[i].grad is not None for i in rnd], prec=self.prec,
                                               gmaxs=[gmaxs[i] for i in rnd],
                                               wpersist=self.wpersist and all(hasattr(items[i][1].weight, "_hr_tstore_range")
                                                                              for i in rnd))
                    for i, o in zip(rnd, got):
                        xs[i].grad = o
                todo = rest
        self._push(bwd)
        return zs

    # ------------------------------------------------------------------ pooling
    def maxpool2(self, x):
        y = Act(ops.maxpool2_fwd(x.data))
        if self.record:
            def bwd():
                g = y.grad
                y.grad = None
                if x.grad is None:
                    x.grad = ops.maxpool2_bwd(x.data, g)
                else:
                    ops.maxpool2_bwd(x.data, g, dx=x.grad, accumulate=True)
            self._push(bwd)
        return y

    # ------------------------------------------------------------------ UNet `up`: upsample x2, pad, concat
    def up_concat(self, low, skip, align_corners=True):
        B, Hs, Ws, Cs = skip.data.shape
        _, Hl, Wl, Cl = low.data.shape
        Hr, Wr = 2 * Hl, 2 * Wl
        py, px = (Hs - Hr) // 2, (Ws - Wr) // 2
        buf = ops.empty_nhwc(B, Hs, Ws, Cs + Cl, skip.data)
        ops.copy(skip.data, buf[..., :Cs])
        ops.bilinear_fwd(low.data, buf[..., Cs:], Hr, Wr, py, px, align_corners)
        cat = Act(buf)
        if self.record:
            def bwd():
                g = cat.grad
                cat.grad = None
                if skip.grad is None:
                    skip.grad = torch.empty(skip.data.shape, dtype=torch.float32, device=g.device)
                    ops.copy(g[..., :Cs], skip.grad)
                else:
                    ops.copy(g[..., :Cs], skip.grad, accumulate=True)
                if low.grad is None:
                    low.grad = ops.bilinear_bwd(g[..., Cs:], low.data.shape, Hr, Wr, py, px, align_corners)
                else:
                    ops.bilinear_bwd(g[..., Cs:], low.data.shape, Hr, Wr, py, px, align_corners, din=low.grad,
                                     accumulate=True)
            self._push(bwd)
        return cat

    # ------------------------------------------------------------------ HRNet fuse: relu(sum of terms)
    def fuse_sum(self, terms, align_corners=True):
        """terms: list of (Act, is_lowres); same-res terms are added, low-res ones bilinearly
        upsampled into the sum; ReLU on the result (Models/models.py:527-542)."""
        same = [a for a, low in terms if not low]
        lows = [a for a, low in terms if low]
        ref = same[0]
        B, H, W, Cn = ref.data.shape
        if len(same) <= 4 and len(lows) <= 3:
            out = ops.fuse_sum([a.data for a in same], [a.data for a in lows], relu=True, align_corners=align_corners)
        else:                                   # (more terms than the one-pass kernel takes: the launch chain)
            if len(same) >= 2:
                out = ops.add(same[0].data, same[1].data, relu=(len(same) == 2 and not lows))
                for i, a in enumerate(same[2:]):
                    ops.add(out, a.data, relu=(i == len(same) - 3 and not lows), out=out)
            else:
                out = torch.empty(ref.data.shape, dtype=torch.float32, device=ref.data.device)
                ops.copy(ref.data, out)
            for i, a in enumerate(lows):
                ops.bilinear_fwd(a.data, out, H, W, 0, 0, align_corners, accumulate=True, relu=(i == len(lows) - 1))
        y = Act(out)
        if self.record:
            def bwd():
                g = ops.relu_bwd(y.grad, y.data, out=y.grad)
                y.grad = None
                for a in lows:
                    if a.grad is None:
                        a.grad = ops.bilinear_bwd(g, a.data.shape, H, W, 0, 0, align_corners)
                    else:
                        ops.bilinear_bwd(g, a.data.shape, H, W, 0, 0, align_corners, din=a.grad, accumulate=True)
                for i, a in enumerate(same):
                    if a.grad is not None:
                        ops.copy(g, a.grad, accumulate=True)
                    elif i == len(same) - 1:
                        a.grad = g                      # last reader takes the buffer itself
                    else:
                        a.grad = torch.empty(g.shape, dtype=torch.float32, device=g.device)
                        ops.copy(g, a.grad)
            self._push(bwd)
        return y

    # ------------------------------------------------------------------ HRNet head concat (models.py:743-747)
    def upsample_concat(self, xs, align_corners=True):
        B, H, W, _ = xs[0].data.shape
        chans = [x.data.shape[3] for x in xs]
        buf = ops.empty_nhwc(B, H, W, sum(chans), xs[0].data)
        ops.copy(xs[0].data, buf[..., :chans[0]])
        off = chans[0]
        for x, c in zip(xs[1:], chans[1:]):
            ops.bilinear_fwd(x.data, buf[..., off:off + c], H, W, 0, 0, align_corners)
            off += c
        cat = Act(buf)
        if self.record:
            def bwd():
                g = cat.grad
                cat.grad = None
                o = 0
                for i, (x, c) in enumerate(zip(xs, chans)):
                    gs = g[..., o:o + c]
                    if i == 0:
                        if x.grad is None:
                            x.grad = torch.empty(x.data.shape, dtype=torch.float32, device=g.device)
                            ops.copy(gs, x.grad)
                        else:
                            ops.copy(gs, x.grad, accumulate=True)
                    elif x.grad is None:
                        x.grad = ops.bilinear_bwd(gs, x.data.shape, H, W, 0, 0, align_corners)
                    else:
                        ops.bilinear_bwd(gs, x.data.shape, H, W, 0, 0, align_corners, din=x.grad, accumulate=True)
                    o += c
            self._push(bwd)
        return cat
