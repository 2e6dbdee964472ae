"""Functional layer over the C ABI: torch tensors in, torch tensors out.

Activations are NHWC fp32 tensors ``[B,H,W,C]`` whose last dim is contiguous;
a channel slice ``t[..., a:b]`` of a wider NHWC tensor is accepted wherever a
row stride ("ld") exists in the ABI.  Every function launches on torch's
current stream and never synchronises.  torch is used for allocation only.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvShape, call, ptr


def _ld(t):
    assert t.dim() == 4 and (t.stride(3) == 1 or t.shape[3] == 1), "NHWC tensor with contiguous channels expected"
    B, H, W, Cn = t.shape
    # strides of size-1 dims are arbitrary: take ld from the first dim that has extent
    if W > 1:
        ld = t.stride(2)
    elif H > 1:
        ld = t.stride(1)
    elif B > 1:
        ld = t.stride(0)
    else:
        ld = Cn
    assert (H == 1 or t.stride(1) == W * ld) and (B == 1 or t.stride(0) == H * W * ld) and ld >= Cn, \
        "only channel-sliced NHWC views are supported"
    return ld


def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


def zeros(shape, dtype, device):
    """zero-filled tensor through the library's fill KERNEL (torch.zeros issues a memset, which a
    captured hipGraph replays out of order on this stack -- see tools/graph_bisect.py)"""
    t = torch.empty(shape, dtype=dtype, device=device)
    flat = t.view(-1).view(torch.float32)      # int64/float64 zeros are all-zero bits too
    call("hrseg_fill", ptr(flat), 0.0, flat.numel())
    return t


def clone(t):
    """device copy through the library's copy KERNEL (clone() of a contiguous tensor is a D2D memcpy
    node under graph capture; kept out of captured graphs like the memsets)"""
    t = t.contiguous()
    out = torch.empty_like(t)
    n = t.numel()
    if t.dtype == torch.float32 and n % 4 == 0 and n > 0:
        call("hrseg_copy", ptr(t), 4, ptr(out), 4, 0, n // 4, 4)
    else:
        out.copy_(t)
    return out


def empty_nhwc(B, H, W, Cn, like):
    return torch.empty((B, H, W, Cn), dtype=torch.float32, device=like.device)


def conv_out_size(h, k, s):
    pad = (k - 1) // 2
    return (h + 2 * pad - k) // s + 1


def _shape(x_shape, ldx, Cout, ldy, k, s, prec=0, gmax=None, residual=None, relu=False, stat=None, wpersist=False,
           x_split=False):
    """gmax: device scalar max|dy| of the gradient operand (fp16x2 data / weight gradients), see hrseg.h;
    residual / relu: the forward's fused epilogue y = relu?(conv + bias + residual) (inference with folded BatchNorm);
    stat: (device buffer of 256*2*Cout doubles, ctypes int) -- BatchNorm partial sums of the output from the epilogue"""
    B, Hi, Wi, Cin = x_shape
    sh = ConvShape(B=B, Hi=Hi, Wi=Wi, Cin=Cin, ldx=ldx, Ho=conv_out_size(Hi, k, s), Wo=conv_out_size(Wi, k, s),
                   Cout=Cout, ldy=ldy, ksize=k, stride=s, precision=prec, grad_absmax=ptr(gmax),
                   residual=ptr(residual), ldr=_ld(residual) if residual is not None else 0, relu=int(bool(relu)))
    if stat is not None:
        sh.stat_partial = ptr(stat[0])
        sh.stat_rows = C.pointer(stat[1])
    sh.w_persistent = int(bool(wpersist))
    sh.x_split = int(bool(x_split))        # x is stored pre-split (bn_fwd_group item "z_split"): hrseg.h
    return sh


def conv_x_split_ok(x_shapes, couts, k, s, prec):
    """may the producer of these convolutions' inputs (NHWC shapes, contiguous) store them pre-split?  (hrseg_conv_x_split_ok:
    every problem on the wave-specialised forward kernels and the nine-tap weight gradient)"""
    shapes = _shape_array([_shape(tuple(xs), xs[3], co, co, k, s, prec) for xs, co in zip(x_shapes, couts)])
    return _lib.conv_x_split_ok(shapes, len(x_shapes))


STAT_ROWS_MAX = 256          # include/hrseg.h hrseg_conv_shape_t.stat_partial: rows the caller provides


def _stat_buffer(cout, device):
    """(partial-sum buffer for the epilogue statistics of one convolution output, host int that receives the row count)"""
    return torch.empty(STAT_ROWS_MAX * 2 * cout, dtype=torch.float64, device=device), C.c_int(0)


# ------------------------------------------------------------------ convolution
_F16_CODES = (_lib.CONV_PRECISION["auto"], _lib.CONV_PRECISION["fp16x2"])
FP16X2_ACT_LIMIT = 65504.0


def absmax(x):
    """max|x| of an NHWC tensor as a 0-dim device tensor (NaN counts as +Inf): hrseg_absmax"""
    slots = zeros((64,), torch.float32, x.device)
    call("hrseg_absmax", ptr(x), _ld(x), _npix(x), x.shape[3], ptr(slots))
    return slots.max()


def _guard_prec(x, prec):
    """fp16x2 convolutions take their activation operand unscaled (include/hrseg.h): exact up to |x| = 65504, Inf / NaN
    results far beyond.  In deterministic mode (the verification mode: it may synchronise) the operand is range-checked
    and an out-of-range or non-finite tensor runs the exact-fp32 kernels instead; counted in `range_fallbacks`."""
    global range_fallbacks
    if not _lib.deterministic() or prec not in _F16_CODES or x.shape[3] % 4 or torch.cuda.is_current_stream_capturing() \
            or _lib.taping():
        return prec            # (a captured graph cannot read a value back: the guard is an eager-mode check)
    if float(absmax(x)) <= FP16X2_ACT_LIMIT:
        return prec
    range_fallbacks += 1
    return _lib.CONV_PRECISION["f32"]


range_fallbacks = 0


def conv_fwd(x, w, bias, k, s, out=None, cout=None, prec=0, residual=None, relu=False, stats=False, wpersist=False,
             x_split=False):
    """x NHWC, w storage [Cout][k*k][Cin] (a channels_last [Cout,Cin,k,k] parameter);
    pass `cout` when w is the flat 1-D parameter slot.  `prec`: _lib.CONV_PRECISION code (all conv functions).
    stats=True: -> (out, (partial, rows) or None): the BatchNorm partial sums of the output where the kernel that ran
    produces them in its epilogue (hrseg_conv_shape_t.stat_partial), else None"""
    _lib.ensure_scratch(x.device)
    if not x_split:                        # (a pre-split tensor is not fp32 data: its producer's fp32 values were in range)
        prec = _guard_prec(x, prec)
    B, Hi, Wi, Cin = x.shape
    Cout = cout if cout is not None else w.shape[0]
    if out is None:
        out = empty_nhwc(B, conv_out_size(Hi, k, s), conv_out_size(Wi, k, s), Cout, x)
    st = _stat_buffer(Cout, x.device) if stats else None
    sh = _shape(x.shape, _ld(x), Cout, _ld(out), k, s, prec, residual=residual, relu=relu, stat=st, wpersist=wpersist,
                x_split=x_split)
    call("hrseg_conv_fwd", ptr(x), ptr(w), ptr(bias), ptr(out), C.byref(sh))
    if stats:
        return out, ((st[0], st[1].value) if st[1].value > 0 else None)
    return out


def conv_dgrad(dy, wt, x_shape, k, s, out=None, accumulate=False, prec=0, gmax=None, wpersist=False):
    """wt = weight_transpose(w): [Cin][k*k][Cout]."""
    _lib.ensure_scratch(dy.device)
    B, Hi, Wi, Cin = x_shape
    if out is None:
        out = empty_nhwc(B, Hi, Wi, Cin, dy)
        accumulate = False
    sh = _shape(x_shape, _ld(out), dy.shape[3], _ld(dy), k, s, prec, gmax, wpersist=wpersist)
    call("hrseg_conv_dgrad", ptr(dy), ptr(wt), ptr(out), int(accumulate), C.byref(sh))
    return out


def conv_wgrad(x, dy, dw, k, s, prec=0, gmax=None, x_split=False):
    """dw (+)= ; dw is the running gradient buffer [Cout][k*k][Cin]."""
    if not x_split:
        prec = _guard_prec(x, prec)
    if prec and k == 3 and s == 1:
        return conv_wgrad_group([x], [dy], [dw], k, s, prec, [gmax], x_splits=[x_split])
    sh = _shape(x.shape, _ld(x), dy.shape[3], _ld(dy), k, s, prec, gmax, x_split=x_split)
    call("hrseg_conv_wgrad", ptr(x), ptr(dy), ptr(dw), C.byref(sh))


def _shape_array(shapes):
    return (ConvShape * len(shapes))(*shapes)


def conv_fwd_group(xs, ws, biases, k, s, couts, prec=0, residuals=None, relus=None, stats=False, wpersist=False, x_splits=None):
    """n independent convolutions (same k, s) in one launch when the library can group them; residuals / relus: the fused
    epilogue per problem (inference with folded BatchNorm); stats=True: -> (outs, [(partial, rows) or None per problem])"""
    _lib.ensure_scratch(xs[0].device)
    if _lib.deterministic() and any(_guard_prec(x, prec) != prec for i, x in enumerate(xs) if not (x_splits and x_splits[i])):
        prec = _lib.CONV_PRECISION["f32"]        # (one precision per grouped call)
    outs, shapes, sts = [], [], []
    for i, (x, co) in enumerate(zip(xs, couts)):
        B, Hi, Wi, Cin = x.shape
        y = empty_nhwc(B, conv_out_size(Hi, k, s), conv_out_size(Wi, k, s), co, x)
        outs.append(y)
        sts.append(_stat_buffer(co, x.device) if stats else None)
        shapes.append(_shape(x.shape, _ld(x), co, _ld(y), k, s, prec, residual=residuals[i] if residuals is not None else None,
                             relu=relus[i] if relus is not None else False, stat=sts[i], wpersist=wpersist,
                             x_split=bool(x_splits[i]) if x_splits else False))
    has_bias = any(b is not None for b in biases)
    call("hrseg_conv_fwd_group", len(xs), _lib.ptr_array(xs), _lib.ptr_array(ws),
         _lib.ptr_array(biases) if has_bias else None, _lib.ptr_array(outs), _shape_array(shapes))
    if stats:
        return outs, [(st[0], st[1].value) if st[1].value > 0 else None for st in sts]
    return outs


def conv_dgrad_group(dys, wts, x_shapes, k, s, outs, accumulate, prec=0, gmaxs=None, wpersist=False):
    """outs[i] None -> allocated (accumulate ignored)"""
    _lib.ensure_scratch(dys[0].device)
    outs, acc, shapes = list(outs), list(accumulate), []
    for i, (dy, xs) in enumerate(zip(dys, x_shapes)):
        if outs[i] is None:
            outs[i] = empty_nhwc(xs[0], xs[1], xs[2], xs[3], dy)
            acc[i] = False
        shapes.append(_shape(xs, _ld(outs[i]), dy.shape[3], _ld(dy), k, s, prec, gmaxs[i] if gmaxs is not None else None,
                             wpersist=wpersist))
    call("hrseg_conv_dgrad_group", len(dys), _lib.ptr_array(dys), _lib.ptr_array(wts), _lib.ptr_array(outs),
         _lib.int_array([int(a) for a in acc]), _shape_array(shapes))
    return outs


def conv_wgrad_group(xs, dys, dws, k, s, prec=0, gmaxs=None, x_splits=None):
    if _lib.deterministic() and any(_guard_prec(x, prec) != prec for i, x in enumerate(xs) if not (x_splits and x_splits[i])):
        prec = _lib.CONV_PRECISION["f32"]
    gm = gmaxs if gmaxs is not None else [None] * len(xs)
    sp = x_splits if x_splits else [False] * len(xs)
    shapes = _shape_array([_shape(x.shape, _ld(x), dy.shape[3], _ld(dy), k, s, prec, g, x_split=bool(q))
                           for x, dy, g, q in zip(xs, dys, gm, sp)])
    nbytes = _lib.conv_wgrad_workspace_bytes(shapes, len(xs)) if (prec and k == 3 and s == 1) else 0
    if nbytes:
        # split-precision 3x3 stride-1 problems: per-block partial sums in a workspace + ordered reduce (no atomics)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=xs[0].device)
        call("hrseg_conv_wgrad_group_ws", len(xs), _lib.ptr_array(xs), _lib.ptr_array(dys), _lib.ptr_array(dws), shapes,
             ptr(ws), nbytes)
        return
    call("hrseg_conv_wgrad_group", len(xs), _lib.ptr_array(xs), _lib.ptr_array(dys), _lib.ptr_array(dws), shapes)


def weight_transpose(w_store, Cout, taps, Cin, out=None):
    if out is None:
        out = torch.empty(Cin * taps * Cout, dtype=torch.float32, device=w_store.device)
    call("hrseg_weight_transpose", ptr(w_store), ptr(out), Cout, taps, Cin)
    return out


# ------------------------------------------------------------------ batch norm
_BN_MAX_CHUNKS = int(os.environ.get("HRSEG_BN_MAX_CHUNKS", "256"))     # pixel chunks (= blocks) per BatchNorm reduction


def _nchunks(npix, Cn):
    q = Cn // 4
    p = 1 if q >= 256 else 256 // q
    return max(1, min(_BN_MAX_CHUNKS, npix // (8 * p)))


def bn_train_coef(y, gamma, beta, running_mean, running_var, nbt, momentum, eps):
    """batch statistics of y -> coef [4][C] (mean, rstd, scale, shift); updates running stats."""
    Cn, npix = y.shape[3], _npix(y)
    nch = _nchunks(npix, Cn)
    part = torch.empty(nch * 2 * Cn, dtype=torch.float64, device=y.device)
    coef = torch.empty(4 * Cn, dtype=torch.float32, device=y.device)
    call("hrseg_bn_stats", ptr(y), _ld(y), npix, Cn, ptr(part), nch)
    call("hrseg_bn_finalize", ptr(part), nch, npix, Cn, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
         ptr(nbt), float(momentum), float(eps), ptr(coef))
    return coef


def bn_fold(w, bias, gamma, beta, running_mean, running_var, eps, cout):
    """BatchNorm (running statistics) folded into the convolution in front of it -> (folded weight [same storage layout],
    folded bias [Cout]); hrseg_bn_fold"""
    w_f = torch.empty_like(w)
    b_f = torch.empty(cout, dtype=torch.float32, device=w.device)
    call("hrseg_bn_fold", ptr(w), ptr(bias), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), cout,
         w.numel() // cout, ptr(w_f), ptr(b_f))
    return w_f, b_f


def bn_eval_coef(gamma, beta, running_mean, running_var, eps):
    Cn = running_mean.numel()
    coef = torch.empty(4 * Cn, dtype=torch.float32, device=running_mean.device)
    call("hrseg_bn_eval_coef", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), Cn, ptr(coef))
    return coef


def bn_apply(y, coef, residual=None, relu=False, out=None):
    if out is None:
        out = torch.empty(y.shape, dtype=torch.float32, device=y.device)
    call("hrseg_bn_apply", ptr(y), _ld(y), ptr(coef), ptr(residual), _ld(residual) if residual is not None else 0,
         int(relu), ptr(out), _ld(out), _npix(y), y.shape[3])
    return out


def bn_bwd(dz, z, relu, y, coef, dgamma, dbeta, dres=None, dres_accumulate=False, eval_mode=False, dy_out=None):
    """-> dy (gradient w.r.t. the conv output); dgamma/dbeta accumulate; dres (+)= g."""
    Cn, npix = y.shape[3], _npix(y)
    nch = _nchunks(npix, Cn)
    part = torch.empty((nch + 1) * 2 * Cn, dtype=torch.float64, device=y.device)   # + totals [2][C]
    dy = dy_out if dy_out is not None else torch.empty(y.shape, dtype=torch.float32, device=y.device)
    zz = z if relu else None
    call("hrseg_bn_bwd_reduce", ptr(dz), _ld(dz), ptr(zz), _ld(z) if relu else 0, int(relu), ptr(y), _ld(y), ptr(coef),
         npix, Cn, ptr(part), nch)
    call("hrseg_bn_bwd_apply", ptr(part), nch, ptr(dz), _ld(dz), ptr(zz), _ld(z) if relu else 0, int(relu), ptr(y),
         _ld(y), ptr(coef), None, ptr(dgamma), ptr(dbeta), ptr(dy), _ld(dy), ptr(dres),
         _ld(dres) if dres is not None else 0, int(dres_accumulate), npix, Cn, int(eval_mode))
    return dy


def _sync_world(sync):
    import torch.distributed as dist
    return dist.get_world_size(sync)


_SYNC_BN_CHECKED = set()


def _check_sync_bn_shapes(sync, items, device):
    """synchronised BatchNorm all-reduces per-chunk partial sums and divides by stat_ranks * npix: every rank must bring the
    same pixel counts (same per-rank batch: `drop_last=True` in the loader).  Checked ONCE per distinct shape signature (one
    tiny all-reduce + readback), then cached: a ragged last batch raises here instead of hanging in a mismatched collective
    or weighting the statistics wrongly."""
    import torch.distributed as dist
    sig = (id(sync), tuple(_npix(it["y"]) for it in items))
    if sig in _SYNC_BN_CHECKED:
        return
    v = torch.tensor([len(items)] + [n for n in sig[1]], dtype=torch.int64, device=device)
    v = torch.cat([v, -v])
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=sync)
    h = v.tolist()
    k = len(h) // 2
    if any(h[i] != -h[k + i] for i in range(k)):
        raise RuntimeError("hrseg_amd sync_bn: the ranks bring different batch / image sizes to a synchronised BatchNorm "
                           f"(max over ranks {h[1:k]}, min {[-x for x in h[k + 1:]]}); use equal per-rank batches (drop_last=True)")
    _SYNC_BN_CHECKED.add(sig)


def bn_fwd_group(items, training, sync=None, phases=7):
    """items: list of dict(y, gamma, beta, rm, rv, nbt, momentum, eps, residual, relu[, out]);
    -> [(z, coef)] with three launches for the whole list (statistics, finalize, apply).
    sync (a process group, training only): cross-rank batch statistics -- the partial sums of all problems live in one
    buffer that is all-reduced between the statistics and the finalize phase (opt-in synchronised BN)."""
    n = len(items)
    arr = (_lib.BnFwd * n)()
    outs = []
    sync = sync if training else None
    ranks = _sync_world(sync) if sync is not None else 1
    pool, pool_off = None, 0
    if sync is not None:
        _check_sync_bn_shapes(sync, items, items[0]["y"].device)
        total = sum(_nchunks(_npix(it["y"]), it["y"].shape[3]) * 2 * it["y"].shape[3] for it in items)
        pool = torch.empty(total, dtype=torch.float64, device=items[0]["y"].device)
    for i, (a, it) in enumerate(zip(arr, items)):
        y = it["y"]
        Cn, npix = y.shape[3], _npix(y)
        z = it.get("out")
        if z is None:
            z = torch.empty(y.shape, dtype=torch.float32, device=y.device)
        coef = torch.empty(4 * Cn, dtype=torch.float32, device=y.device)
        res = it.get("residual")
        a.y, a.ldy, a.npix, a.C = ptr(y), _ld(y), npix, Cn
        a.gamma, a.beta = ptr(it["gamma"]), ptr(it["beta"])
        a.running_mean, a.running_var = ptr(it["rm"]), ptr(it["rv"])
        a.num_batches_tracked = ptr(it["nbt"]) if training else None
        a.momentum, a.eps = float(it["momentum"]), float(it["eps"])
        a.stat_updates = int(it.get("repeat", 1))
        a.stat_div = int(it.get("stat_div", 1))
        a.residual, a.ldr, a.relu = ptr(res), (_ld(res) if res is not None else 0), int(it["relu"])
        a.z, a.ldz, a.coef = ptr(z), _ld(z), ptr(coef)
        a.relu_mask = ptr(it.get("relu_mask"))
        a.z_split = int(bool(it.get("z_split", False)))
        a.residual_split = int(bool(it.get("residual_split", False)))
        a.stat_ranks = ranks
        if training and it.get("partial") is not None:
            part, nch = it["partial"]                   # partial sums left by the convolution's epilogue (phases 6)
            a.partial, a.nchunks = ptr(part), int(nch)
            outs.append((z, coef, part))
        elif training:
            nch = _nchunks(npix, Cn)
            if pool is not None:
                part = pool[pool_off:pool_off + nch * 2 * Cn]
                pool_off += nch * 2 * Cn
            else:
                part = torch.empty(nch * 2 * Cn, dtype=torch.float64, device=y.device)
            a.partial, a.nchunks = ptr(part), nch
            outs.append((z, coef, part))
        else:
            a.partial, a.nchunks = None, 0
            outs.append((z, coef, None))
    if sync is not None:
        import torch.distributed as dist
        call("hrseg_bn_fwd_group_phases", n, arr, 1, 1)
        dist.all_reduce(pool, op=dist.ReduceOp.SUM, group=sync)
        call("hrseg_bn_fwd_group_phases", n, arr, 1, 6)
    elif phases != 7:           # (measurement switches of engine.py only)
        call("hrseg_bn_fwd_group_phases", n, arr, int(training), int(phases))
    else:
        call("hrseg_bn_fwd_group", n, arr, int(training))
    return [(z, coef) for z, coef, _ in outs]


def bn_bwd_group(items, eval_mode, sync=None):
    """items: list of dict(dz, z, relu, y, coef, dgamma, dbeta, dres, dres_accumulate); dy is written
    in place over dz.  Three launches for the whole list.  sync: see bn_fwd_group (the backward's batch means become
    global; dgamma / dbeta receive this rank's share, so the gradient all-reduce sums them to the global value)."""
    n = len(items)
    arr = (_lib.BnBwd * n)()
    keep = []
    sync = None if eval_mode else sync
    ranks = _sync_world(sync) if sync is not None else 1
    pool, pool_off = None, 0
    if sync is not None:
        total = 0
        for it in items:
            Cn, npix, nseg = it["y"].shape[3], _npix(it["y"]), int(it.get("nseg", 1))
            nch = _nchunks(npix, Cn)
            nch = max(nseg, nch // nseg * nseg) if nseg > 1 else nch
            total += (nch + nseg) * 2 * Cn
        pool = zeros((total,), torch.float64, items[0]["y"].device)
    for i, (a, it) in enumerate(zip(arr, items)):
        y, dz, z = it["y"], it["dz"], it["z"]
        Cn, npix = y.shape[3], _npix(y)
        nseg = int(it.get("nseg", 1))
        nch = _nchunks(npix, Cn)
        if nseg > 1:                                  # chunks never straddle two segments
            nch = max(nseg, nch // nseg * nseg)
        if pool is not None:
            part = pool[pool_off:pool_off + (nch + nseg) * 2 * Cn]
            pool_off += (nch + nseg) * 2 * Cn
        else:
            part = torch.empty((nch + nseg) * 2 * Cn, dtype=torch.float64, device=y.device)
        keep.append(part)
        a.nseg = nseg
        a.sum_ranks = ranks
        dres = it.get("dres")
        a.dz, a.lddz = ptr(dz), _ld(dz)
        use_z = it["relu"] and z is not None      # z None: mask recomputed from y (forward without residual)
        a.z, a.ldz, a.relu = (ptr(z) if use_z else None), (_ld(z) if use_z else 0), int(it["relu"])
        a.y, a.ldy, a.coef = ptr(y), _ld(y), ptr(it["coef"])
        a.dgamma, a.dbeta = ptr(it["dgamma"]), ptr(it["dbeta"])
        a.dy, a.lddy = ptr(dz), _ld(dz)
        a.dres, a.lddres = ptr(dres), (_ld(dres) if dres is not None else 0)
        a.dres_accumulate = int(bool(it.get("dres_accumulate", False)))
        a.npix, a.C, a.partial, a.nchunks = npix, Cn, ptr(part), nch
        a.dy_absmax = ptr(it.get("dy_absmax"))
        a.relu_mask = ptr(it.get("relu_mask"))
    if sync is not None:
        import torch.distributed as dist
        call("hrseg_bn_bwd_group_phases", n, arr, 0, 1)
        dist.all_reduce(pool, op=dist.ReduceOp.SUM, group=sync)
        call("hrseg_bn_bwd_group_phases", n, arr, 0, 6)
    else:
        call("hrseg_bn_bwd_group", n, arr, int(eval_mode))
    return [it["dz"] for it in items]


# ------------------------------------------------------------------ pooling / resize / glue
def maxpool2_fwd(x):
    B, H, W, Cn = x.shape
    y = empty_nhwc(B, H // 2, W // 2, Cn, x)
    call("hrseg_maxpool2_fwd", ptr(x), _ld(x), ptr(y), _ld(y), B, H, W, Cn)
    return y


def maxpool2_bwd(x, dy, dx=None, accumulate=False):
    B, H, W, Cn = x.shape
    if dx is None:
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        accumulate = False
    call("hrseg_maxpool2_bwd", ptr(x), _ld(x), ptr(dy), _ld(dy), ptr(dx), _ld(dx), int(accumulate), B, H, W, Cn)
    return dx


def bilinear_fwd(x, out, Hr, Wr, py=0, px=0, align_corners=True, accumulate=False, relu=False):
    """resize x to Hr x Wr and place it at (py,px) of `out` (NHWC view, may be a channel slice)."""
    B, Hi, Wi, Cn = x.shape
    call("hrseg_bilinear_fwd", ptr(x), _ld(x), B, Hi, Wi, Cn, ptr(out), _ld(out), out.shape[1], out.shape[2], Hr, Wr,
         py, px, int(align_corners), int(accumulate), int(relu))
    return out


def bilinear_bwd(dout, x_shape, Hr, Wr, py=0, px=0, align_corners=True, din=None, accumulate=False):
    B, Hi, Wi, Cn = x_shape
    if din is None:
        din = empty_nhwc(B, Hi, Wi, Cn, dout)
        accumulate = False
    call("hrseg_bilinear_bwd", ptr(dout), _ld(dout), B, Hi, Wi, Cn, ptr(din), _ld(din), dout.shape[1], dout.shape[2],
         Hr, Wr, py, px, int(align_corners), int(accumulate))
    return din


def fuse_sum(same, lows, relu=True, align_corners=True):
    """relu?(sum of same-resolution NHWC tensors + sum of bilinearly up-sampled low-resolution ones) in one pass"""
    ref = same[0]
    B, H, W, Cn = ref.shape
    out = torch.empty(ref.shape, dtype=torch.float32, device=ref.device)
    call("hrseg_fuse_sum", len(same), _lib.ptr_array(same), _lib.int_array([_ld(t) for t in same]), len(lows),
         _lib.ptr_array(lows) if lows else None, _lib.int_array([_ld(t) for t in lows]) if lows else None,
         _lib.int_array([t.shape[1] for t in lows]) if lows else None, _lib.int_array([t.shape[2] for t in lows]) if lows else None,
         ptr(out), _ld(out), B, H, W, Cn, int(align_corners), int(relu))
    return out


def add(a, b, relu=False, out=None):
    if out is None:
        out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    call("hrseg_add", ptr(a), _ld(a), ptr(b), _ld(b), ptr(out), _ld(out), int(relu), _npix(a), a.shape[3])
    return out


def copy(src, dst, accumulate=False):
    call("hrseg_copy", ptr(src), _ld(src), ptr(dst), _ld(dst), int(accumulate), _npix(src), src.shape[3])
    return dst


def relu_bwd(dz, z, out=None):
    if out is None:
        out = torch.empty(z.shape, dtype=torch.float32, device=z.device)
    call("hrseg_relu_bwd", ptr(dz), _ld(dz), ptr(z), _ld(z), ptr(out), _ld(out), _npix(z), z.shape[3])
    return out


def nchw_to_nhwc(x, out=None):
    B, Cn, H, W = x.shape
    if out is None:
        out = empty_nhwc(B, H, W, Cn, x)
    call("hrseg_nchw_to_nhwc", ptr(x), ptr(out), Cn, B, Cn, H, W)
    return out


def nhwc_to_nchw(x):
    B, H, W, Cn = x.shape
    out = torch.empty((B, Cn, H, W), dtype=torch.float32, device=x.device)
    call("hrseg_nhwc_to_nchw", ptr(x), _ld(x), ptr(out), B, Cn, H, W)
    return out


def concat_image_logits(x, z):
    """NCHW image [B,Ci,H,W] + NCHW logits [B,C,H,W] -> NHWC [B,H,W,Ci+C] (the input of a logit-concatenated level pass)"""
    x, z = _c(x.float()), _c(z.float())
    B, Ci, H, W = x.shape
    Cz = z.shape[1]
    out = torch.empty((B, H, W, Ci + Cz), dtype=torch.float32, device=x.device)
    call("hrseg_nchw_to_nhwc", ptr(x), ptr(out), Ci + Cz, B, Ci, H, W)
    call("hrseg_nchw_to_nhwc", ptr(z), out.data_ptr() + 4 * Ci, Ci + Cz, B, Cz, H, W)
    return out


def nhwc_slice_to_nchw(g, c0):
    """channels [c0:] of an NHWC tensor as a contiguous NCHW tensor"""
    B, H, W, Cn = g.shape
    assert g.is_contiguous()
    out = torch.empty((B, Cn - c0, H, W), dtype=torch.float32, device=g.device)
    call("hrseg_nhwc_to_nchw", g.data_ptr() + 4 * c0, Cn, ptr(out), B, Cn - c0, H, W)
    return out


def accumulate_flat(dst, src):
    """dst += src for two contiguous fp32 tensors of the same size (the copy kernel's accumulate form)"""
    n = dst.numel()
    assert src.numel() == n and dst.is_contiguous() and src.is_contiguous()
    if n % 4 == 0 and n > 0:
        call("hrseg_copy", ptr(src), 4, ptr(dst), 4, 1, n // 4, 4)
    else:
        dst.add_(src)
    return dst


def encode_targets(label, on_lut, parent):
    """label [B,H,W] uint8 (device), on_lut [256] int64 bit table (device), parent list[int] -> [B,C,H,W] fp32"""
    assert label.dtype == torch.uint8 and label.is_cuda and label.dim() == 3
    assert on_lut.dtype == torch.int64 and on_lut.numel() == 256 and on_lut.is_cuda
    label = label.contiguous()
    B, H, W = label.shape
    Cn = len(parent)
    out = torch.empty((B, Cn, H, W), dtype=torch.float32, device=label.device)
    call("hrseg_encode_targets", ptr(label), ptr(on_lut), _lib.int_array(list(parent)), ptr(out), B, Cn, H * W)
    return out


def combine_levels(x0, x1, masks, is_union):
    """x0 [B,C0,H,W] (+ x1 [B,C1,H,W] or None), per output channel a bit mask over the C0+C1 input channels and a
    union flag -> [B,len(masks),H,W]: copy of the selected channel, or 1.0 where any selected channel is > 0"""
    x0 = _c(x0.float())
    B, C0, H, W = x0.shape
    C1 = 0
    if x1 is not None:
        x1 = _c(x1.float())
        C1 = x1.shape[1]
    out = torch.empty((B, len(masks), H, W), dtype=torch.float32, device=x0.device)
    call("hrseg_combine_levels", ptr(x0), C0, ptr(x1), C1, (C.c_ulonglong * len(masks))(*[int(m) for m in masks]),
         _lib.int_array([int(u) for u in is_union]), ptr(out), B, len(masks), H * W)
    return out


def fill(t, v):
    call("hrseg_fill", ptr(t), float(v), t.numel())
    return t


# ------------------------------------------------------------------ FiLM / head / composition
def gap_nchw(p):
    B, Cn, H, W = p.shape
    cond = torch.empty((B, Cn), dtype=torch.float32, device=p.device)
    scratch = torch.empty(B * Cn * 64, dtype=torch.float64, device=p.device)
    call("hrseg_gap_nchw", ptr(p), ptr(cond), ptr(scratch), B * Cn, H * W)
    return cond


def film_linear_fwd(cond, wl, bl):
    B, Cc = cond.shape
    F2 = bl.numel()
    gb = torch.empty((B, F2), dtype=torch.float32, device=cond.device)
    call("hrseg_film_linear_fwd", ptr(cond), ptr(wl), ptr(bl), ptr(gb), B, Cc, F2)
    return gb


def film_linear_bwd(cond, wl, dgb, dwl, dbl, dcond_scale, want_dcond=True):
    B, Cc = cond.shape
    dcond = torch.empty((B, Cc), dtype=torch.float32, device=cond.device) if want_dcond else None
    call("hrseg_film_linear_bwd", ptr(cond), ptr(wl), ptr(dgb), ptr(dcond), ptr(dwl), ptr(dbl), B, Cc, dgb.shape[1],
         float(dcond_scale))
    return dcond


def head_fwd(f, gb, w, bias, cout=None):
    """f NHWC [B,H,W,F]; w [Cout,F] storage; -> z NHWC [B,H,W,Cout]"""
    B, H, W, F = f.shape
    Cout = cout if cout is not None else w.shape[0]
    z = empty_nhwc(B, H, W, Cout, f)
    call("hrseg_head_fwd", ptr(f), _ld(f), ptr(gb), ptr(w), ptr(bias), ptr(z), Cout, B, H * W, F, Cout)
    return z


def head_bwd(f, gb, w, dz, dw, dbias, dgb, want_df=True, df=None, df_accumulate=False, cout=None):
    B, H, W, F = f.shape
    Cout = cout if cout is not None else w.shape[0]
    if want_df and df is None:
        df = torch.empty(f.shape, dtype=torch.float32, device=f.device)
        df_accumulate = False
    call("hrseg_head_bwd", ptr(f), _ld(f), ptr(gb), ptr(w), ptr(dz), _ld(dz), ptr(df), _ld(df) if df is not None else 0,
         int(df_accumulate), ptr(dw), ptr(dbias), ptr(dgb), B, H * W, F, Cout)
    return df


def logits_up_fwd(z, Ho, Wo, align_corners=True):
    B, Hi, Wi, Cn = z.shape
    out = torch.empty((B, Cn, Ho, Wo), dtype=torch.float32, device=z.device)
    call("hrseg_logits_up_fwd", ptr(z), _ld(z), B, Hi, Wi, Cn, ptr(out), Ho, Wo, int(align_corners))
    return out


def logits_up_bwd(dout, Hi, Wi, align_corners=True):
    B, Cn, Ho, Wo = dout.shape
    din = empty_nhwc(B, Hi, Wi, Cn, dout)
    call("hrseg_logits_up_bwd", ptr(dout), B, Hi, Wi, Cn, ptr(din), Cn, Ho, Wo, int(align_corners))
    return din


def sigmoid_fwd(z):
    p = torch.empty_like(z)
    call("hrseg_sigmoid_fwd", ptr(z), ptr(p), z.numel())
    return p


def _strides3(dp, B, Cn, hw):
    """(sb, sc, si) element strides of a [B,C,H,W]-shaped (possibly expanded) gradient."""
    if dp.dim() == 4:
        assert dp.stride(3) in (0, 1) and (dp.stride(2) == dp.shape[3] * dp.stride(3) or dp.shape[2] == 1)
        return dp.stride(0), dp.stride(1), dp.stride(3)
    assert dp.dim() == 2
    return dp.stride(0), dp.stride(1), 0


def sigmoid_bwd(dp, z, dz=None, accumulate=False):
    B, Cn, H, W = z.shape
    if dz is None:
        dz = torch.empty_like(z)
        accumulate = False
    sb, sc, si = _strides3(dp, B, Cn, H * W)
    call("hrseg_sigmoid_bwd", ptr(dp), sb, sc, si, ptr(z), ptr(dz), int(accumulate), B, Cn, H * W)
    return dz


def compose_fwd(z, pprev, group_parent, group_size):
    B, Cn, H, W = z.shape
    p = torch.empty_like(z)
    call("hrseg_compose_fwd", ptr(z), ptr(pprev), ptr(p), B, Cn, pprev.shape[1], H * W, len(group_parent),
         _lib.int_array(group_parent), _lib.int_array(group_size))
    return p


def compose_bwd(dp, z, pprev, group_parent, group_size, dz=None, dz_accumulate=False, dpprev=None,
                dpprev_accumulate=False, want_dz=True, want_dpprev=True):
    B, Cn, H, W = z.shape
    if want_dz and dz is None:
        dz, dz_accumulate = torch.empty_like(z), False
    if want_dpprev and dpprev is None:
        dpprev, dpprev_accumulate = torch.empty_like(pprev), False
    sb, sc, si = _strides3(dp, B, Cn, H * W)
    call("hrseg_compose_bwd", ptr(dp), sb, sc, si, ptr(z), ptr(pprev), ptr(dz), int(dz_accumulate), ptr(dpprev),
         int(dpprev_accumulate), B, Cn, pprev.shape[1], H * W, len(group_parent), _lib.int_array(group_parent),
         _lib.int_array(group_size))
    return dz, dpprev


# ------------------------------------------------------------------ loss / metrics / optimizer
def _c(t):
    """NCHW planes are addressed as b*C*hw + c*hw + i: insist on contiguous storage"""
    return t if t.is_contiguous() else t.contiguous()


def loss_fwd(z, t, w):
    """-> (out[3] = ce, dice, n_valid_dice ; coef for the backward)"""
    z, t = _c(z), _c(t)
    B, Cn, H, W = z.shape
    part = torch.empty(B * Cn * 5, dtype=torch.float64, device=z.device)
    out = torch.empty(3, dtype=torch.float32, device=z.device)
    coef = torch.empty(B * Cn * 4, dtype=torch.float32, device=z.device)
    call("hrseg_loss_partials", ptr(z), ptr(t), ptr(part), B, Cn, H * W)
    call("hrseg_loss_finalize", ptr(part), ptr(w), B, Cn, ptr(out), ptr(coef))
    return out, coef


def loss_bwd(z, t, coef, g, dz=None, accumulate=False):
    z, t = _c(z), _c(t)
    B, Cn, H, W = z.shape
    if dz is None:
        dz, accumulate = torch.empty_like(z), False
    call("hrseg_loss_bwd", ptr(z), ptr(t), ptr(coef), ptr(g), ptr(dz), int(accumulate), B, Cn, H * W)
    return dz


def consistency_sums(p, pprev, group_parent, group_size):
    """-> [ngroups] float64 sums of |sum_children P - P_parent| over batch and pixels"""
    p, pprev = _c(p), _c(pprev)
    B, Cn, H, W = p.shape
    out = zeros((len(group_parent),), torch.float64, p.device)
    call("hrseg_consistency", ptr(p), ptr(pprev), ptr(out), B, Cn, pprev.shape[1], H * W, len(group_parent),
         _lib.int_array(group_parent), _lib.int_array(group_size))
    return out


def consistency_bwd(p, pprev, g, scale, group_parent, group_size):
    """-> (dp, dpprev) for  scale * g * sum |sum_children P - P_parent|"""
    p, pprev = _c(p), _c(pprev)
    B, Cn, H, W = p.shape
    dp, dpprev = torch.empty_like(p), torch.empty_like(pprev)
    call("hrseg_consistency_bwd", ptr(p), ptr(pprev), ptr(g), float(scale), ptr(dp), ptr(dpprev), B, Cn, pprev.shape[1],
         H * W, len(group_parent), _lib.int_array(group_parent), _lib.int_array(group_size))
    return dp, dpprev


def group_kl_sums(z, pprev, group_parent, group_size):
    """-> [ngroups] float64: sum over b, pixels and the group's children of Q (log Q + log size) (hrseg_group_kl)"""
    z, pprev = _c(z), _c(pprev)
    B, Cn, H, W = z.shape
    out = zeros((len(group_parent),), torch.float64, z.device)
    call("hrseg_group_kl", ptr(z), ptr(pprev), ptr(out), B, Cn, pprev.shape[1], H * W, len(group_parent),
         _lib.int_array(group_parent), _lib.int_array(group_size))
    return out


def group_kl_bwd(z, pprev, g, scale, group_parent, group_size):
    z, pprev = _c(z), _c(pprev)
    B, Cn, H, W = z.shape
    dz = torch.empty_like(z)
    call("hrseg_group_kl_bwd", ptr(z), ptr(pprev), ptr(g), float(scale), ptr(dz), B, Cn, pprev.shape[1], H * W,
         len(group_parent), _lib.int_array(group_parent), _lib.int_array(group_size))
    return dz


def predict_metrics(z, t, child, mask_pred=True, want_onehot=True):
    """-> (onehot or None, confusion matrix [K,K] int64 with K = C + child)"""
    z, t = _c(z), _c(t)
    B, Cn, H, W = z.shape
    K = Cn + (1 if child else 0)
    onehot = torch.empty_like(z) if (want_onehot and mask_pred) else None
    cm = zeros((K, K), torch.int64, z.device)
    call("hrseg_predict_metrics", ptr(z), ptr(t), ptr(onehot), ptr(cm), B, Cn, H * W, int(child), int(mask_pred))
    return onehot, cm


def metric_vectors(cms, child):
    """per-level confusion matrices [K,K] int64 (predict_metrics) -> [5][sum C_L] fp32: accuracy, iou, dice, precision,
    recall of every class, the levels side by side; one launch (hrseg_metric_vectors)"""
    cms = [_c(cm) for cm in cms]
    K = [cm.shape[0] for cm in cms]
    total = sum(k - (1 if ch else 0) for k, ch in zip(K, child))
    out = torch.empty((5, total), dtype=torch.float32, device=cms[0].device)
    call("hrseg_metric_vectors", len(cms), _lib.ptr_array(cms), _lib.int_array(K), _lib.int_array([int(bool(c)) for c in child]),
         ptr(out))
    return out


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, gscale=1.0):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    call("hrseg_adamw", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
         float(wd), float(bc1), float(bc2), float(gscale))


def adamw_dev(p, g, m, v, hyper, state):
    """graph-replayable AdamW step: hyper/state are device tensors (see hrseg.h)"""
    call("hrseg_adamw_dev", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(hyper), ptr(state))
