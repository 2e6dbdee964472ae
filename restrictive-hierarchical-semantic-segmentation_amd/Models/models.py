"""UNet / HRNet-W48 donors with the hierarchical level loop, on the MI355X engine.

Call surface and state_dict layout of the reference's Models/models.py:
``UNet(size, n_channels, hierarchy, model_type)``, ``HighResolutionNet(config,
hierarchy, model_type)``, ``FiLM``, ``get_level_classes``,
``build_hierarchy_indices``; ``forward`` returns ``([], logits)`` for flat
models and ``(probs_per_level, logits_per_level)`` (NCHW fp32) otherwise.

Every FLOP runs in the HIP kernels of csrc/ through engine.Recorder; the
torch.nn classes below only hold parameters under the reference's names
(models.py line numbers cited per class).  torch.autograd sees one node per
model call (_Bridge), so the reference loop's ``loss.backward()`` drives the
engine's reverse pass.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import _lib, ops
from ..engine import Act, FlatParams, Recorder
from ..utils.hierarchy import build_hierarchy_indices, child_groups, get_level_classes  # noqa: F401 (API)

BN_MOMENTUM = 0.1
DEFAULT_CONV_DTYPE = "auto"


# ----------------------------------------------------------------------------- parameter holders
class Conv2d(nn.Conv2d):
    """parameter holder; the convolution itself is hrseg_conv_* (csrc/conv.hip)"""

    def forward(self, x):
        raise RuntimeError("hrseg_amd layers run inside the model's engine, not standalone")


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        raise RuntimeError("hrseg_amd layers run inside the model's engine, not standalone")


class FiLM(nn.Module):
    """reference models.py:58-77: gamma,beta = Linear(GAP(cond_map)); feats*gamma+beta.
    Inside the models the modulation is folded into the 1x1 head kernel."""

    def __init__(self, feat_ch: int, cond_ch: int):
        super().__init__()
        self.cond_pool = nn.AdaptiveAvgPool2d(1)
        self.mlp = nn.Sequential(nn.Flatten(), nn.Linear(cond_ch, 2 * feat_ch))
        self.feat_ch = feat_ch

    def gamma_beta(self, cond_map):
        """[B,2F] from a probability map (NCHW) or a ready [B,Cc] vector"""
        lin = self.mlp[1]
        cond = ops.gap_nchw(cond_map.contiguous()) if cond_map.dim() == 4 else cond_map.contiguous()
        return ops.film_linear_fwd(cond, _store(lin.weight), _store(lin.bias))

    def forward(self, feats, cond_map):
        """standalone FiLM on an NCHW tensor (inference utility, no autograd)"""
        _lib.require_gpu()
        with torch.no_grad():
            gb = self.gamma_beta(cond_map)
            f = feats.shape[1]
            return feats * gb[:, :f, None, None] + gb[:, f:, None, None]


SMALL_CIN_MAX = 8       # csrc/conv_common.h HRSEG_SMALL_CIN_MAX: widest input of the direct (non-MFMA) first-layer kernels


def _check_cond_stem_width(widths):
    """concat_prev_logits: level L's first convolution reads cat(image, logits_{L-1}); it runs the direct Cin <= 8 kernels"""
    for L, w in enumerate(widths, start=1):
        if w > SMALL_CIN_MAX:
            raise ValueError(f"concat_prev_logits: level {L} would encode {w} input channels (image + the logits of level "
                             f"{L - 1}); the first-layer kernels take at most {SMALL_CIN_MAX}")


def _store(p):
    return p._hr_store if hasattr(p, "_hr_store") else p.detach().contiguous()


# ----------------------------------------------------------------------------- autograd bridge
class _Bridge(torch.autograd.Function):
    """One autograd node for a whole model call."""

    @staticmethod
    def forward(ctx, anchor, model, x, record):
        run = model._run(x, record)
        ctx.run = run if record else None
        ctx.set_materialize_grads(False)
        # return ALIASES: autograd stamps its grad_fn on the returned objects, and grad_fn -> ctx.run ->
        # run.probs would otherwise close a reference cycle through the C++ node that no collector breaks
        # (the step then retains its logits/probabilities for ever)
        outs = tuple(t.detach() for t in run.probs) + tuple(t.detach() for t in run.logits)
        ctx.n_probs = len(run.probs)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        run = ctx.run
        if run is None or run.done:
            raise RuntimeError("hrseg_amd: backward through a model call twice (activations were freed)")
        run.backward(list(grads[:ctx.n_probs]), list(grads[ctx.n_probs:]))
        return None, None, None, None


class _Run:
    """State of one model call: per-level recorders and what the level loop saved."""

    def __init__(self, model):
        self.model = model
        self.levels = []        # per level dict
        self.probs, self.logits = [], []
        self.done = False
        self.shared_tape = False   # batched / de-duplicated passes: every level's head hangs off ONE backbone tape
        self.batched_feats = None  # batched passes: the stacked [L*B,...] feature map the heads slice

    def backward(self, dprobs, dlogits):
        m = self.model
        flat = m._flat
        flat.prepare_backward()
        n = len(self.levels)
        hierarchical = len(self.probs) > 0
        # gradient w.r.t. P_L: full tensors (dp_full) and broadcast [B,C] parts (dp_bcast)
        dp_full = [dprobs[L] if hierarchical and dprobs[L] is not None else None for L in range(n)]
        dp_own = [False] * n            # True once the buffer is ours to accumulate into
        dp_bcast = [None] * n
        dz_input = [None] * n           # logit-concatenated re-encoding: d loss / d logits_L through level L+1's input
        for L in reversed(range(n)):
            lv = self.levels[L]
            z = lv["z"]
            dz, own = dlogits[L], False
            if dz is not None:
                dz = dz.contiguous()
            if dz_input[L] is not None:
                dz, own = (dz_input[L], True) if dz is None else (ops.accumulate_flat(ops.clone(dz), dz_input[L]), True)
            if hierarchical:
                for dp in (dp_full[L], dp_bcast[L]):
                    if dp is None:
                        continue
                    if dz is not None and not own:
                        dz, own = ops.clone(dz), True  # never write into autograd's tensor
                    if L == 0:
                        dz = ops.sigmoid_bwd(dp, z, dz=dz, accumulate=dz is not None)
                        own = True
                    elif lv["groups"]:
                        gp, gs = lv["groups"]
                        prev = dp_full[L - 1]
                        if prev is not None and not dp_own[L - 1]:
                            prev = ops.clone(prev)
                        dz, prev = ops.compose_bwd(dp, z, self.probs[L - 1], gp, gs, dz=dz,
                                                   dz_accumulate=dz is not None, dpprev=prev,
                                                   dpprev_accumulate=prev is not None)
                        dp_full[L - 1], dp_own[L - 1], own = prev, True, True
            if dz is None:
                # nothing flows into this level's logits: its pass contributes no gradient
                if not self.shared_tape:
                    lv["rec"].tape.clear()
                continue
            dcond = m._head_backward(lv, dz)
            if dcond is not None and L > 0:
                dp_bcast[L - 1] = dcond if dp_bcast[L - 1] is None else dp_bcast[L - 1] + dcond
            if self.shared_tape:
                continue                 # the heads' feature gradients accumulate; one reverse pass below
            hook = m._grad_hook if (L == 0) else None
            lv["rec"].backward(hook)
            xin = lv.get("xin")
            if xin is not None and xin.grad is not None and L > 0:
                dz_input[L - 1] = ops.nhwc_slice_to_nchw(xin.grad, xin.grad.shape[3] - self.logits[L - 1].shape[1])
            lv.clear()
        if self.shared_tape and n > 0:
            lv0 = self.levels[0]
            root = self.batched_feats if self.batched_feats is not None else lv0["feats"]
            if root.grad is not None:
                lv0["rec"].backward(m._grad_hook)
            else:
                lv0["rec"].tape.clear()
            self.batched_feats = None
            for lv in self.levels:
                lv.clear()
        # weight gradients ran on the side stream: everything after the reverse pass (all-reduce tail,
        # optimizer) is ordered behind them
        from ..engine import wgrad_stream
        side = wgrad_stream(flat.grad.device)
        if side is not None:
            _lib.stream_wait(None, side)
        _lib.tape_release()
        flat.attach_grads()
        if m._grad_hook is not None:
            _lib.host_call(lambda: m._grad_hook("end"))
        self.levels = []
        self.probs, self.logits = [], []
        self.done = True


class _EngineModel(nn.Module):
    """Shared level loop (reference models.py:257-306 / :751-802) over an engine backbone."""

    align_corners = True
    batch_passes_by_default = True

    def _init_engine(self):
        self._flat = None
        self._anchor = None
        self._grad_hook = None       # DDP: called with tape marks during the last backward level
        # opt-in: run the L identical level passes of a hierarchical model once (see _run)
        self.dedup_passes = os.environ.get("HRSEG_DEDUP_PASSES", "0") == "1"
        # run the L training passes one after the other (as the reference does) instead of batched
        # (default per model: HRNet's many small layers gain 12 % from batching, UNet's few large ones nothing)
        # arithmetic of the convolution contractions (include/hrseg.h hrseg_conv_precision): "f32" = exact fp32 MFMA,
        # "bf16x3" = fp32 operands split into three bf16 pieces (fp32-grade results on the bf16 matrix pipe),
        # "bf16x2" / "bf16" = explicit reduced-precision opt-ins (BASELINE configs[4]: bf16 inputs, fp32 accumulate)
        self.conv_dtype = os.environ.get("HRSEG_CONV_DTYPE", DEFAULT_CONV_DTYPE)
        env = os.environ.get("HRSEG_SEQUENTIAL_PASSES")
        self.sequential_passes = (env == "1") if env is not None else not self.batch_passes_by_default
        # opt-in extensions, all default off: `concat_prev_logits` is a constructor argument (it adds the
        # cond_stems parameters); `sync_bn` = BatchNorm statistics over all ranks (what the reference's SyncBatchNorm would
        # do WITH a process group, bn_helper.py:4-11 -- the reference itself never synchronises)
        self.sync_bn = os.environ.get("HRSEG_SYNC_BN", "0") == "1"
        self.sync_bn_group = None
        # inference (eval mode, nothing recorded): BatchNorm folded into the convolution weights, residual + ReLU in the
        # convolution epilogue (hrseg_bn_fold, hrseg_conv_shape_t.residual / relu).  HRSEG_BN_FOLD=0: separate BN launches
        self.fold_bn = os.environ.get("HRSEG_BN_FOLD", "1") != "0"
        self._param_epoch = 0            # bumped whenever weights or running statistics may have changed

    # -- parameters -------------------------------------------------------------------------
    def flatten_parameters(self, device=None):
        device = device or next(self.parameters()).device
        if self._flat is None or not self._flat.valid(device):
            self._flat = FlatParams(self, device)
            for b in self.buffers():
                if b.device != device:
                    raise RuntimeError("hrseg_amd: parameters and buffers live on different devices")
        return self._flat

    # -- public forward ---------------------------------------------------------------------
    def _forward_impl(self, x):
        _lib.require_gpu()
        if x.device.type != "cuda":
            raise RuntimeError("hrseg_amd models run on the GPU only: move the model and the input to 'cuda'")
        self.flatten_parameters(x.device)
        if self._anchor is None or self._anchor.device != x.device:
            self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
        outs = _Bridge.apply(self._anchor, self, x, torch.is_grad_enabled())
        n_levels = len(self.levels) if self._hier() else 1
        if not self._hier():
            return [], outs[0]
        return list(outs[:n_levels]), list(outs[n_levels:])

    def _hier(self):
        return self.model_type != 0

    # -- engine forward ---------------------------------------------------------------------
    def _run(self, x, record):
        run = _Run(self)
        prec = _lib.CONV_PRECISION[self.conv_dtype]
        if self.training:
            self._param_epoch += 1       # running statistics move
        fold = self._fold() if (not self.training and not record) else None
        self._flat.wt_stale = True          # weights may have been updated since the last call
        wpersist = self._flat.refresh_weight_images(record)     # one launch: every cached weight image for the current weights
        x = x.contiguous().float()
        x_nhwc = None                       # built where a pass needs it (the batched passes build their own stack)
        size = (x.shape[2], x.shape[3])
        if not self._hier():
            rec = Recorder(self.training, record, self._flat, prec=prec, sync=self._bn_sync(), fold=fold, wpersist=wpersist)
            feats = self._backbone(rec, Act(ops.nchw_to_nhwc(x), needs_grad=False))
            z, lv = self._head_forward(rec, feats, self._flat_head(), None, None, size)
            lv.update(rec=rec, groups=None)
            run.levels.append(lv)
            run.logits.append(z)
            return run
        # The reference re-runs the backbone on the image for every level (D1): L bit-identical passes.
        # Default in training: faithful re-execution (BN statistics update L times, gradients sum over
        # the L tapes).  In inference (eval mode, nothing recorded) the features are computed once.
        # `dedup_passes` (explicit opt-in, SURVEY.md "dedup mode") does the same in training: ONE pass
        # whose BN layers apply their running-stat update L times, the L heads read the shared features,
        # their feature gradients are summed and ONE reverse pass runs -- the same result up to fp32
        # summation order at 1/L of the backbone work (executed FLOPs change; bench.py reports it apart).
        n_levels = len(self.levels)
        concat = bool(getattr(self, "concat_prev_logits", False))
        if concat and self.dedup_passes:
            raise RuntimeError("hrseg_amd: dedup_passes needs identical level passes; concat_prev_logits makes every level "
                               "re-encode its own input")
        dedup = bool(self.dedup_passes) and self.training and n_levels > 1
        # Default in training: the L passes run BATCHED -- the image batch is stacked L times and every layer
        # is one launch for all passes.  Each pass is still computed in full (same FLOPs as L sequential
        # passes, same values: batch statistics of L identical copies are those of one pass, the unbiased
        # variance uses one pass's pixel count, running statistics take L updates, the backward normalises
        # each pass on its own); it halves the launch count and doubles the work per launch.
        # `sequential_passes` (HRSEG_SEQUENTIAL_PASSES=1) runs them one after the other as the reference does.
        batched = (self.training and n_levels > 1 and not dedup and not self.sequential_passes and not concat
                   # 32-bit pixel indices in the kernels: larger stacks run the passes sequentially
                   and n_levels * x.shape[0] * x.shape[2] * x.shape[3] < (1 << 31))
        run.shared_tape = (dedup or batched) and record
        shared, shared_rec = None, None
        Bn = x.shape[0]
        for L in range(n_levels):
            xin = None
            if batched:
                if shared is None:
                    rec = Recorder(self.training, record, self._flat, bn_repeat=n_levels, bn_segments=n_levels, prec=prec,
                                   sync=self._bn_sync(), wpersist=wpersist)
                    # the image batch stacked L times (library copy kernels: no ATen launch on the path, so a launch
                    # tape of the step is complete)
                    stack = torch.empty((n_levels * Bn, x.shape[2], x.shape[3], x.shape[1]), dtype=torch.float32,
                                        device=x.device)
                    for rep in range(n_levels):
                        ops.nchw_to_nhwc(x, out=stack[rep * Bn:(rep + 1) * Bn])
                    xx = Act(stack, needs_grad=False)
                    shared, shared_rec = self._backbone(rec, xx), rec
                    run.batched_feats = shared
                feats, rec = Act(shared.data[L * Bn:(L + 1) * Bn]), shared_rec
                feats.slot = L               # its gradient is rows [L*B, (L+1)*B) of the stacked feature gradient
            elif shared is None:
                rec = Recorder(self.training, record, self._flat, bn_repeat=n_levels if dedup else 1, prec=prec,
                               sync=self._bn_sync(), fold=fold, wpersist=wpersist)
                if concat and L > 0:
                    # level L re-encodes cat(image, logits_{L-1}) through its own first convolution (cond_stems[L-1])
                    xin = Act(ops.concat_image_logits(x, run.logits[L - 1]), needs_grad=record)
                    feats = self._backbone(rec, xin, first=self.cond_stems[L - 1])
                else:
                    if x_nhwc is None:
                        x_nhwc = Act(ops.nchw_to_nhwc(x), needs_grad=False)
                    feats = self._backbone(rec, x_nhwc)
                if (dedup or (not self.training and not record)) and not concat:
                    shared, shared_rec = feats, rec
            else:
                feats, rec = shared, shared_rec
            film = self.films[L - 1] if L > 0 else None
            z, lv = self._head_forward(rec, feats, self._level_head(L), film, run.probs[L - 1] if L > 0 else None, size)
            groups = None
            if L == 0:
                p = ops.sigmoid_fwd(z)
            else:
                g = self.child_groups[L - 1]
                if len(g) == 0:
                    p = ops.zeros(z.shape, torch.float32, z.device)
                else:
                    gp = [self.levels[L - 1].index(pname) for pname, _ in g]
                    gs = [len(ch) for _, ch in g]
                    groups = (gp, gs)
                    p = ops.compose_fwd(z, run.probs[L - 1], gp, gs)
            lv.update(rec=rec, groups=groups, stacked=run.batched_feats if batched else None,
                      xin=xin)
            run.levels.append(lv)
            run.probs.append(p)
            run.logits.append(z)
        return run

    def _fold(self):
        """(conv, bn) -> folded (weight, bias) for the inference path, cached on the conv until the parameters or the running
        statistics may have changed (training forward, optimizer step, load_state_dict, or an in-place edit of the flat
        buffer that torch's version counter sees)"""
        if not self.fold_bn:
            return None
        token = (self._param_epoch, self._flat.data._version)

        def fold(conv, bn):
            hit = getattr(conv, "_hr_fold", None)
            tok = token + (bn.running_mean._version, bn.running_var._version)
            if hit is None or hit[0] != tok:
                w_f, b_f = ops.bn_fold(conv.weight._hr_store, conv.bias._hr_store if conv.bias is not None else None,
                                       bn.weight._hr_store, bn.bias._hr_store, bn.running_mean, bn.running_var, bn.eps,
                                       conv.out_channels)
                # fp16x2 scales weights by 2^8 before the split (include/hrseg.h): a folded weight beyond ~255 (a tiny running
                # variance under a large gamma) would leave the fp16 range.  Checked ONCE per fold (one readback per layer when
                # the parameters changed, not per forward); such a layer runs the exact-fp32 kernels.
                n4 = w_f.numel() // 4 * 4
                wmax = float(ops.absmax(w_f[:n4].view(1, 1, n4 // 4, 4))) if n4 else 0.0
                hit = conv._hr_fold = (tok, w_f, b_f, wmax <= 240.0)
            return hit[1], hit[2], hit[3]
        return fold

    def notify_parameters_changed(self):
        """an optimizer (or anyone writing parameters / buffers outside torch's in-place ops) calls this: folded inference
        weights are rebuilt on the next eval forward"""
        self._param_epoch += 1

    def load_state_dict(self, *args, **kwargs):
        self._param_epoch += 1
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        """a train <-> eval switch also invalidates the folded inference weights: whatever ran in train mode (an eager
        step, a hipGraph / launch-tape replay that moves weights and running statistics through raw pointers, an EMA swap
        through .data) is then seen by the next eval forward without anyone having to call notify_parameters_changed()"""
        if hasattr(self, "_param_epoch") and bool(mode) != self.training:
            self._param_epoch += 1
        return super().train(mode)

    def _bn_sync(self):
        """process group for cross-rank BatchNorm statistics, or None (the default: statistics stay per rank)"""
        import torch.distributed as dist
        if not self.sync_bn or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.sync_bn_group) == 1:
            return None
        return self.sync_bn_group if self.sync_bn_group is not None else dist.group.WORLD

    def _head_forward(self, rec, feats, head, film, p_prev, size):
        """FiLM (folded) + 1x1 head (+ bilinear resize for HRNet) -> logits NCHW."""
        w, b = head.weight, head.bias
        cond = gb = None
        if film is not None:
            lin = film.mlp[1]
            cond = ops.gap_nchw(p_prev)
            gb = ops.film_linear_fwd(cond, lin.weight._hr_store, lin.bias._hr_store)
        zl = ops.head_fwd(feats.data, gb, w._hr_store, b._hr_store, cout=head.out_channels)
        if (zl.shape[1], zl.shape[2]) != tuple(size):
            z = ops.logits_up_fwd(zl, size[0], size[1], self.align_corners)
        else:
            z = ops.nhwc_to_nchw(zl)
        lv = dict(feats=feats, head=head, film=film, cond=cond, gb=gb, z=z, low=(zl.shape[1], zl.shape[2]),
                  hw=size[0] * size[1])
        return z, lv

    def _head_backward(self, lv, dz):
        """dz (NCHW, full res) -> gradients of head / FiLM; seeds feats.grad; returns the
        broadcast gradient [B,Cprev] w.r.t. the previous level's probabilities (or None)."""
        feats, head, film = lv["feats"], lv["head"], lv["film"]
        B, Cn, H, W = dz.shape
        if lv["low"] != (H, W):
            dzl = ops.logits_up_bwd(dz, lv["low"][0], lv["low"][1], self.align_corners)
        else:
            dzl = ops.nchw_to_nhwc(dz)
        dgb = ops.zeros(lv["gb"].shape, torch.float32, lv["gb"].device) if film is not None else None
        stacked = lv.get("stacked")
        if stacked is not None:          # batched passes: this level's rows of the stacked feature gradient
            if stacked.grad is None:
                stacked.grad = ops.zeros(stacked.data.shape, torch.float32, stacked.data.device)
            Bn = feats.data.shape[0]
            ops.head_bwd(feats.data, lv["gb"], head.weight._hr_store, dzl, head.weight._hr_gstore, head.bias._hr_gstore,
                         dgb, df=stacked.grad[feats.slot * Bn:(feats.slot + 1) * Bn], cout=head.out_channels)
        else:
            g = ops.head_bwd(feats.data, lv["gb"], head.weight._hr_store, dzl, head.weight._hr_gstore,
                             head.bias._hr_gstore, dgb, cout=head.out_channels)
            if feats.grad is None:
                feats.grad = g
            else:                        # shared features (de-duplicated passes): sum over the levels' heads
                ops.add(feats.grad, g, out=feats.grad)
        if film is None:
            return None
        lin = film.mlp[1]
        return ops.film_linear_bwd(lv["cond"], lin.weight._hr_store, dgb, lin.weight._hr_gstore,
                                   lin.bias._hr_gstore, 1.0 / lv["hw"])


# ----------------------------------------------------------------------------- UNet (models.py:108-306)
class double_conv(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Sequential(Conv2d(in_ch, out_ch, 3, padding=1), BatchNorm2d(out_ch), nn.ReLU(inplace=True),
                                  Conv2d(out_ch, out_ch, 3, padding=1), BatchNorm2d(out_ch), nn.ReLU(inplace=True))

    def run(self, rec, x, first=None):
        x = rec.conv_bn(x, first if first is not None else self.conv[0], self.conv[1], relu=True, split_for=self.conv[3])
        return rec.conv_bn(x, self.conv[3], self.conv[4], relu=True)


class inconv(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = double_conv(in_ch, out_ch)

    def run(self, rec, x, first=None):
        return self.conv.run(rec, x, first)


class down(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), double_conv(in_ch, out_ch))

    def run(self, rec, x):
        return self.mpconv[1].run(rec, rec.maxpool2(x))


class up(nn.Module):
    def __init__(self, in_ch, out_ch, bilinear=True):
        super().__init__()
        if not bilinear:
            raise NotImplementedError("the reference only ever builds up(bilinear=True)")
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = double_conv(in_ch, out_ch)

    def run(self, rec, x1, x2):
        return self.conv.run(rec, rec.up_concat(x1, x2))


class outconv(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = Conv2d(in_ch, out_ch, 1)


class UNet(_EngineModel):
    """Flat (type==0): returns [], logits.  Hierarchical (type==1): level loop with FiLM."""
    batch_passes_by_default = False     # measured: 84.8 ms batched vs 83.2 ms sequential (hier, 620x620, B=4)

    def __init__(self, size=620, n_channels=1, hierarchy={}, model_type=0, concat_prev_logits=False):
        """concat_prev_logits (opt-in extension; north_star: "backbone re-run with logit-concatenated input";
        the reference re-runs on the image only, models.py:267,277): level L >= 1 re-encodes cat(image, logits_{L-1})
        through its own first convolution cond_stems[L-1]; every other layer is shared.  The level passes are then no
        longer identical: they run one after the other, and dedup_passes is refused."""
        super().__init__()
        self.model_type = model_type
        self.hierarchy = hierarchy
        self.concat_prev_logits = bool(concat_prev_logits) and model_type != 0
        self.inc0 = inconv(n_channels, 64)
        self.down1, self.down2 = down(64, 128), down(128, 256)
        self.down3, self.down4 = down(256, 512), down(512, 512)
        self.up1, self.up2 = up(1024, 256), up(512, 128)
        self.up3, self.up4 = up(256, 64), up(128, 64)
        if model_type == 0:
            n_leaves = sum(len(v) for v in get_level_classes(hierarchy, inc_parent=False).values())
            self.out_flat = outconv(64, n_leaves)
        else:
            self.levels, self.parent_of, self.children_of = build_hierarchy_indices(hierarchy)
            self.child_groups = child_groups(self.levels, self.children_of)
            self.heads = nn.ModuleList([outconv(64, len(self.levels[0]))])
            for groups in self.child_groups:
                n = sum(len(ch) for _, ch in groups)
                self.heads.append(outconv(64, n if n > 0 else 1))
            self.films = nn.ModuleList([FiLM(feat_ch=64, cond_ch=len(self.levels[L - 1]))
                                        for L in range(1, len(self.levels))])
            if self.concat_prev_logits:
                _check_cond_stem_width([n_channels + self.heads[L - 1].conv.out_channels for L in range(1, len(self.levels))])
                self.cond_stems = nn.ModuleList([Conv2d(n_channels + self.heads[L - 1].conv.out_channels, 64, 3, padding=1)
                                                 for L in range(1, len(self.levels))])
        self._init_engine()

    def _flat_head(self):
        return self.out_flat.conv

    def _level_head(self, L):
        return self.heads[L].conv

    def _backbone(self, rec, x, first=None):
        x1 = self.inc0.run(rec, x, first)
        x2 = self.down1.run(rec, x1)
        rec.mark("down2")            # gradient all-reduce bucket boundaries (parallel.py)
        x3 = self.down2.run(rec, x2)
        x4 = self.down3.run(rec, x3)
        rec.mark("down4")
        x5 = self.down4.run(rec, x4)
        d = self.up1.run(rec, x5, x4)
        rec.mark("up2")
        d = self.up2.run(rec, d, x3)
        d = self.up3.run(rec, d, x2)
        rec.mark("up4")
        return self.up4.run(rec, d, x1)

    def forward(self, x, type=0, hierarchy={}, threshold=0.5):
        if self.model_type != 0 and type == 0:
            # the reference takes the flat branch on type==0 and fails on the missing flat head
            raise AttributeError("'UNet' object has no attribute 'out_flat'")
        return self._forward_impl(x)


# ----------------------------------------------------------------------------- HRNet (models.py:322-832)
def conv3x3(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _bn(c):
    return BatchNorm2d(c, momentum=BN_MOMENTUM)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = conv3x3(inplanes, planes, stride), _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = conv3x3(planes, planes), _bn(planes)
        self.downsample = downsample
        self.stride = stride

    def run(self, rec, x):
        r = x if self.downsample is None else rec.conv_bn(x, self.downsample[0], self.downsample[1], relu=False)
        y = rec.conv_bn(x, self.conv1, self.bn1, relu=True, split_for=self.conv2)        # (conv2 is y's only reader)
        return rec.conv_bn(y, self.conv2, self.bn2, relu=True, residual=r)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = Conv2d(inplanes, planes, kernel_size=1, bias=False), _bn(planes)
        self.conv2, self.bn2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False), _bn(planes)
        self.conv3, self.bn3 = Conv2d(planes, planes * 4, kernel_size=1, bias=False), _bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def run(self, rec, x):
        r = x if self.downsample is None else rec.conv_bn(x, self.downsample[0], self.downsample[1], relu=False)
        y = rec.conv_bn(x, self.conv1, self.bn1, relu=True, split_for=self.conv2)
        y = rec.conv_bn(y, self.conv2, self.bn2, relu=True)
        return rec.conv_bn(y, self.conv3, self.bn3, relu=True, residual=r)


blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _make_layer(block, inplanes, planes, blocks, stride=1):
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(Conv2d(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                   _bn(planes * block.expansion))
    layers = [block(inplanes, planes, stride, downsample)]
    layers += [block(planes * block.expansion, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def _run_seq(rec, seq, x):
    for blk in seq:
        x = blk.run(rec, x)
    return x


def _run_conv_chain(rec, chain, x):
    """Sequential of Sequential(conv, bn[, relu]) (fuse down paths, transitions)"""
    for step in chain:
        x = rec.conv_bn(x, step[0], step[1], relu=len(step) == 3)
    return x


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method=None,
                 multi_scale_output=True, align_corners=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError(f"NUM_BRANCHES({num_branches}) <> NUM_BLOCKS({len(num_blocks)}) / "
                             f"NUM_CHANNELS({len(num_channels)}) / NUM_INCHANNELS({len(num_inchannels)})")
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.align_corners = align_corners
        self.branches = nn.ModuleList()
        for b in range(num_branches):
            self.branches.append(_make_layer(blocks, num_inchannels[b], num_channels[b], num_blocks[b]))
            self.num_inchannels[b] = num_channels[b] * blocks.expansion
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(inplace=True)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        ch, nb = self.num_inchannels, self.num_branches
        rows = []
        for i in range(nb if self.multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(Conv2d(ch[j], ch[i], 1, 1, 0, bias=False), _bn(ch[i])))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        cout = ch[i] if last else ch[j]
                        mods = [Conv2d(ch[j], cout, 3, 2, 1, bias=False), _bn(cout)]
                        if not last:
                            mods.append(nn.ReLU(inplace=True))
                        steps.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def _run_branches(self, rec, xs):
        """The branches are independent chains of equally many BasicBlocks: walk them in lockstep
        so that step k of every branch goes out as one grouped launch (engine.conv_bn_group)."""
        nb = self.num_branches
        depth = len(self.branches[0])
        lockstep = all(len(br) == depth and all(isinstance(blk, BasicBlock) and blk.downsample is None for blk in br)
                       for br in self.branches)
        if not lockstep:
            return [_run_seq(rec, self.branches[b], xs[b]) for b in range(nb)]
        xs = list(xs)
        for kblk in range(depth):
            blks = [self.branches[b][kblk] for b in range(nb)]
            ys = rec.conv_bn_group([(xs[b], blks[b].conv1, blks[b].bn1, None) for b in range(nb)], relu=True,
                                   single_reader=True, split_for=[blks[b].conv2 for b in range(nb)])
            # a block's output is read by the next block of the branch only (its conv1, and its BatchNorm apply as the
            # residual: residual_split) -- except the last block's, which the fuse layers read as fp32
            nxt = [self.branches[b][kblk + 1].conv1 for b in range(nb)] if kblk + 1 < depth else None
            xs = rec.conv_bn_group([(ys[b], blks[b].conv2, blks[b].bn2, xs[b]) for b in range(nb)], relu=True, split_for=nxt,
                                   split_level=2)
        return xs

    def run(self, rec, xs):
        if self.num_branches == 1:
            return [_run_seq(rec, self.branches[0], xs[0])]
        xs = self._run_branches(rec, xs)
        nb = self.num_branches
        # every first step of the fuse paths only needs the branch outputs: ALL 1x1 convs of the up paths are
        # one grouped launch, the stride-2 3x3 convs of the down chains one grouped launch per chain depth
        # (ReLU per item: every conv of a chain but its last).  Several paths read the same branch output;
        # the engine issues their data gradients in rounds of distinct inputs.
        nf = len(self.fuse_layers)
        term = {}
        ups = [(i, j) for i in range(nf) for j in range(nb) if j > i]
        for c0 in range(0, len(ups), 8):
            chunk = ups[c0:c0 + 8]
            outs_ = rec.conv_bn_group([(xs[j], self.fuse_layers[i][j][0], self.fuse_layers[i][j][1], None)
                                       for i, j in chunk], relu=False)
            for key, o in zip(chunk, outs_):
                term[key] = o
        chains = {(i, j): xs[j] for i in range(nf) for j in range(nb) if j < i}
        depth = 0
        while True:
            live = [(i, j) for (i, j) in chains if len(self.fuse_layers[i][j]) > depth]
            if not live:
                break
            for c0 in range(0, len(live), 8):
                chunk = live[c0:c0 + 8]
                outs_ = rec.conv_bn_group([(chains[(i, j)], self.fuse_layers[i][j][depth][0],
                                            self.fuse_layers[i][j][depth][1], None) for i, j in chunk],
                                          relu=[len(self.fuse_layers[i][j][depth]) == 3 for i, j in chunk])
                for key, o in zip(chunk, outs_):
                    chains[key] = o
            depth += 1
        outs = []
        for i in range(len(self.fuse_layers)):
            terms = []
            for j in range(nb):
                if j == i:
                    terms.append((xs[j], False))
                elif j > i:
                    terms.append((term[(i, j)], True))
                else:
                    terms.append((chains[(i, j)], False))
            outs.append(rec.fuse_sum(terms, self.align_corners))
        return outs


class HighResolutionNet(_EngineModel):
    """Hierarchy-aware HRNet (flat: [], logits; hierarchical: probs, logits per level)."""

    def __init__(self, config, hierarchy={}, model_type=0, concat_prev_logits=False, **kwargs):
        super().__init__()
        extra = config.MODEL.EXTRA
        self.align_corners = bool(config.MODEL.ALIGN_CORNERS)
        self.model_type = model_type
        self.concat_prev_logits = bool(concat_prev_logits) and model_type != 0      # see UNet
        self.hierarchy = hierarchy
        self.relu = nn.ReLU(inplace=True)
        self.stem = nn.Sequential(Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False), _bn(64),
                                  nn.ReLU(inplace=True),
                                  Conv2d(64, 64, kernel_size=3, stride=2, padding=1, bias=False), _bn(64),
                                  nn.ReLU(inplace=True))
        self.stage1_cfg = extra["STAGE1"]
        block = blocks_dict[self.stage1_cfg["BLOCK"]]
        self.layer1 = _make_layer(block, 64, self.stage1_cfg["NUM_CHANNELS"][0], self.stage1_cfg["NUM_BLOCKS"][0])
        pre = [block.expansion * self.stage1_cfg["NUM_CHANNELS"][0]]
        for idx in (2, 3, 4):
            cfg = extra[f"STAGE{idx}"]
            setattr(self, f"stage{idx}_cfg", cfg)
            block = blocks_dict[cfg["BLOCK"]]
            chans = [c * block.expansion for c in cfg["NUM_CHANNELS"]]
            setattr(self, f"transition{idx - 1}", self._make_transition_layer(pre, chans))
            stage, pre = self._make_stage(cfg, chans)
            setattr(self, f"stage{idx}", stage)
        last = int(sum(pre))
        self.shared_head = nn.Sequential(Conv2d(last, last, kernel_size=1, stride=1, padding=0, bias=True), _bn(last),
                                         nn.ReLU(inplace=True))
        final_k = extra["FINAL_CONV_KERNEL"]
        if final_k != 1:
            raise NotImplementedError("FINAL_CONV_KERNEL=3 heads are not built (the shipped W48 config uses 1)")
        if model_type == 0:
            n_leaves = sum(len(v) for v in get_level_classes(hierarchy, inc_parent=False).values())
            self.classifier = Conv2d(last, n_leaves, kernel_size=1, stride=1, padding=0)
        else:
            self.levels, self.parent_of, self.children_of = build_hierarchy_indices(hierarchy)
            self.child_groups = child_groups(self.levels, self.children_of)
            self.classifiers = nn.ModuleList([Conv2d(last, len(self.levels[0]), kernel_size=1)])
            for groups in self.child_groups:
                n = sum(len(ch) for _, ch in groups)
                self.classifiers.append(Conv2d(last, n if n > 0 else 1, kernel_size=1))
            self.films = nn.ModuleList([FiLM(feat_ch=last, cond_ch=len(self.levels[L - 1]))
                                        for L in range(1, len(self.levels))])
            if self.concat_prev_logits:
                _check_cond_stem_width([3 + self.classifiers[L - 1].out_channels for L in range(1, len(self.levels))])
                self.cond_stems = nn.ModuleList([Conv2d(3 + self.classifiers[L - 1].out_channels, 64, kernel_size=3, stride=2,
                                                        padding=1, bias=False) for L in range(1, len(self.levels))])
        self._init_engine()

    _make_layer = staticmethod(_make_layer)

    def _make_stage(self, layer_config, num_inchannels, multi_scale_output=True):
        block = blocks_dict[layer_config["BLOCK"]]
        modules = []
        for i in range(layer_config["NUM_MODULES"]):
            multi = multi_scale_output or i != layer_config["NUM_MODULES"] - 1
            modules.append(HighResolutionModule(layer_config["NUM_BRANCHES"], block, layer_config["NUM_BLOCKS"],
                                                num_inchannels, layer_config["NUM_CHANNELS"],
                                                layer_config.get("FUSE_METHOD"), multi, self.align_corners))
            num_inchannels = modules[-1].get_num_inchannels()
        return nn.Sequential(*modules), num_inchannels

    @staticmethod
    def _make_transition_layer(pre, cur):
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                if c != pre[i]:
                    layers.append(nn.Sequential(Conv2d(pre[i], c, 3, 1, 1, bias=False), _bn(c), nn.ReLU(inplace=True)))
                else:
                    layers.append(None)
            else:
                steps = []
                for j in range(i + 1 - len(pre)):
                    cout = c if j == i - len(pre) else pre[-1]
                    steps.append(nn.Sequential(Conv2d(pre[-1], cout, 3, 2, 1, bias=False), _bn(cout),
                                               nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def _flat_head(self):
        return self.classifier

    def _level_head(self, L):
        return self.classifiers[L]

    def _backbone(self, rec, x, first=None):
        # rec.mark(name): "every parameter registered at or after module `name` is final once the
        # reverse pass gets back here" (gradient all-reduce buckets, parallel.py)
        x = rec.conv_bn(x, first if first is not None else self.stem[0], self.stem[1], relu=True)
        x = rec.conv_bn(x, self.stem[3], self.stem[4], relu=True)
        rec.mark("layer1")
        x = _run_seq(rec, self.layer1, x)
        ys = [x]
        for t_idx in (1, 2, 3):
            rec.mark(f"transition{t_idx}")
            cfg = getattr(self, f"stage{t_idx + 1}_cfg")
            trans = getattr(self, f"transition{t_idx}")
            xs = []
            for i in range(cfg["NUM_BRANCHES"]):
                if trans[i] is None:
                    xs.append(ys[i])
                else:
                    src = ys[i] if i < len(ys) else ys[-1]
                    if isinstance(trans[i][0], nn.Sequential):
                        xs.append(_run_conv_chain(rec, trans[i], src))
                    else:
                        xs.append(rec.conv_bn(src, trans[i][0], trans[i][1], relu=True))
            ys = xs
            for mod in getattr(self, f"stage{t_idx + 1}"):
                ys = mod.run(rec, ys)
        rec.mark("shared_head")
        cat = rec.upsample_concat(ys, self.align_corners)
        return rec.conv_bn(cat, self.shared_head[0], self.shared_head[1], relu=True)

    def forward(self, x):
        return self._forward_impl(x)

    def init_weights(self, pretrained="", device="cpu"):
        """suffix/shape matching checkpoint loader (reference models.py:804-832)"""
        checkpoint = torch.load(pretrained, map_location=device)
        if "state_dict" in checkpoint:
            checkpoint = checkpoint["state_dict"]
        stripped = {}
        for k, v in checkpoint.items():
            for prefix in ("model.", "module.", "net.", "network."):
                if k.startswith(prefix):
                    k = k[len(prefix):]
            stripped[k] = v
        own = self.state_dict()
        picked = {}
        for mk, mv in own.items():
            if mk in stripped and stripped[mk].size() == mv.size():
                picked[mk] = stripped[mk]
                continue
            for ck, cv in stripped.items():
                if (mk.endswith(ck) or ck.endswith(mk)) and cv.size() == mv.size():
                    picked[mk] = cv
                    break
        missing = set(own) - set(picked)
        print(f"Loaded {len(picked)} / {len(own)} layers.")
        if missing:
            print(f"Missing {len(missing)} layers (first 10): {list(missing)[:10]}")
        own.update(picked)
        self.load_state_dict(own)
        return self
