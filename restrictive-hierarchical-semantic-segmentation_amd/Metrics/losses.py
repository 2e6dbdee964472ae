"""Class-weighted CE + soft-Dice + hierarchy-consistency losses on the GPU.

Call surface of the reference's Metrics/losses.py: ``CrossEntropyLoss(smooth)``,
``SoftDiceLoss(smooth, num_classes)`` called as ``loss(outs, targets,
logits_input=False, class_weight=None)`` and ``hierarchical_consistency_loss(
probs_per_level, levels, parent_of, reduction='mean')``.

One fused HIP pass over (logits, targets) (hrseg_loss_partials) produces the five
masked per-(b,c) sums both losses need; the reference walks B*C boolean-mask
gathers per loss (losses.py:52-59, :100-114), each a device->host sync.  The
gradient is one more pass (hrseg_loss_bwd).  Semantics kept: mask = target != -1,
CE item -> constant 1.0 when any class mask of the item is empty (:116), Dice
items with 0/0 dropped (:64), Dice -> None when no item survives (:66).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops

_weight_cache = {}


def _weights(class_weight, device):
    if class_weight is None:
        # the reference fails here too (losses.py:56-59 / :103-112)
        raise TypeError("class_weight is required: the reference losses index it per class")
    key = (tuple(float(v) for v in class_weight), str(device))
    w = _weight_cache.get(key)
    if w is None:
        w = torch.tensor(key[0], dtype=torch.float32, device=device)
        _weight_cache[key] = w
    return w


class _FusedLoss(torch.autograd.Function):
    """(logits, targets, w) -> [ce, dice, n_valid_dice_items]"""

    @staticmethod
    def forward(ctx, z, t, w):
        z = z.contiguous()
        t = t.contiguous()
        out, coef = ops.loss_fwd(z, t, w)
        ctx.save_for_backward(z, t, coef)
        return out

    @staticmethod
    def backward(ctx, g):
        z, t, coef = ctx.saved_tensors
        return ops.loss_bwd(z, t, coef, g.contiguous()), None, None


def fused_ce_dice(logits, targets, class_weight):
    """-> tensor [3] = (CE, Dice, number of valid Dice items); differentiable w.r.t. logits."""
    if logits.shape[1] != len(class_weight):
        raise ValueError(f"class_weight has {len(class_weight)} entries, logits have {logits.shape[1]} channels")
    return _FusedLoss.apply(logits, targets, _weights(class_weight, logits.device))


def global_batch_dice(res, group=None):
    """Dice term of one level under data parallelism.  The reference computes Dice on the batch gathered on one GPU:
    mean over the items whose Dice is not 0/0 (losses.py:64-66), so the divisor is the GLOBAL number of valid items.
    A rank-local mean divides by the rank's own count, which differs from rank to rank as soon as one shard holds a
    sample with no valid pixel at this level.  With n_r = this rank's count and n = sum over ranks (ONE all-reduce of
    one float per level, stream-ordered, no host sync) the rank's term becomes

        dice_r * n_r * world / n      (0 when n == 0)

    whose average over the ranks -- what GradSync's summed gradient times AdamW's 1/world evaluates -- is
    sum_b dice_b / n: exactly the gathered-batch Dice and its gradient.  `res` = fused_ce_dice(...)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return res[1]
    n_local = res[2].detach()
    n = n_local.clone().reshape(1)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    world = float(dist.get_world_size(group))
    scale = torch.where(n > 0, n_local.reshape(1) * world / n.clamp(min=1.0), torch.zeros_like(n))
    return res[1] * scale.reshape(())


def _as_logits(outs, logits_input, log_domain):
    """The kernels take logits.  Already-normalised inputs are mapped back:
    log-probabilities are their own logits; probabilities go through log()."""
    if logits_input or log_domain:
        return outs
    return torch.log(outs)


class CrossEntropyLoss(nn.Module):
    def __init__(self, smooth=0.0):
        super().__init__()
        self.smooth = smooth

    def forward(self, outs, targets, logits_input=False, class_weight=None):
        return fused_ce_dice(_as_logits(outs, logits_input, True), targets, class_weight)[0]


class SoftDiceLoss(nn.Module):
    def __init__(self, smooth=0, num_classes=3):
        super().__init__()
        self.smooth = smooth

    def forward(self, outs, targets, logits_input=False, class_weight=None):
        res = fused_ce_dice(_as_logits(outs, logits_input, False), targets, class_weight)
        if float(res[2]) == 0.0:          # every item was 0/0: the reference returns None
            return None
        return res[1]


def _level_groups(levels, parent_of, L):
    """[(parent channel at L-1, [child channels at L])] for parents that have children at L"""
    out = []
    for p_idx, p_name in enumerate(levels[L - 1]):
        ch = [i for i, c in enumerate(levels[L]) if parent_of.get(c, None) == p_name]
        if ch:
            out.append((p_idx, ch))
    return out


class _LevelConsistency(torch.autograd.Function):
    """sum over the level's parent groups of (mean or sum over b,h,w of) |sum_children P - P_parent|"""

    @staticmethod
    def forward(ctx, cur, prev, gp, gs, scale):
        cur, prev = cur.contiguous(), prev.contiguous()
        sums = ops.consistency_sums(cur, prev, gp, gs)
        ctx.save_for_backward(cur, prev)
        ctx.meta = (gp, gs, scale)
        return (sums.sum() * scale).float()

    @staticmethod
    def backward(ctx, g):
        cur, prev = ctx.saved_tensors
        gp, gs, scale = ctx.meta
        dcur, dprev = ops.consistency_bwd(cur, prev, g.reshape(1).float().contiguous(), scale, gp, gs)
        return dcur, dprev, None, None, None


def hierarchical_consistency_loss(probs_per_level, levels, parent_of, reduction="mean"):
    """mean_{b,h,w} |sum_children P_c - P_p| averaged over parents (losses.py:150-177), forward and
    gradient on the GPU (hrseg_consistency, hrseg_consistency_bwd).  In the training loop the inputs
    are one-hot predictions without gradient; probabilities that require grad get theirs."""
    if probs_per_level is None or levels is None or parent_of is None:
        return probs_per_level[0].sum() * 0 if probs_per_level else 0.0
    total, count = None, 0
    for L in range(1, len(levels)):
        groups = _level_groups(levels, parent_of, L)
        if not groups:
            continue
        prev, cur = probs_per_level[L - 1], probs_per_level[L]
        contiguous = all(ch == list(range(ch[0], ch[0] + len(ch))) for _, ch in groups) and \
            [c for _, ch in groups for c in ch] == list(range(cur.shape[1]))
        if not contiguous:
            raise NotImplementedError("children of a parent must occupy consecutive channels (BFS channel order)")
        n = cur.shape[0] * cur.shape[2] * cur.shape[3]
        scale = 1.0 / n if reduction == "mean" else 1.0
        part = _LevelConsistency.apply(cur, prev, [p for p, _ in groups], [len(ch) for _, ch in groups], scale)
        total = part if total is None else total + part
        count += len(groups)
    if count == 0:
        return probs_per_level[0].sum() * 0
    return total / count


class _GroupKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, pprev, gp, gs):
        z, pprev = z.contiguous(), pprev.contiguous()
        sums = ops.group_kl_sums(z, pprev, gp, gs)
        n = z.shape[0] * z.shape[2] * z.shape[3]
        sizes = torch.tensor([float(v) for v in gs], dtype=torch.float64, device=z.device)
        ctx.save_for_backward(z, pprev)
        ctx.meta = (gp, gs, 1.0 / (n * len(gs)))
        return ((sums / sizes).sum() / (n * len(gs))).float()

    @staticmethod
    def backward(ctx, g):
        z, pprev = ctx.saved_tensors
        gp, gs, scale = ctx.meta
        return ops.group_kl_bwd(z, pprev, g.reshape(1).float().contiguous(), scale, gp, gs), None, None, None


def grouped_conditional_kl(z_children_all, probs_prev_level, groups, levels_prev):
    """The optional stabiliser the reference ships commented out (Metrics/losses.py:180-210), same signature: per parent
    group KL(Q_{c|p} || Uniform) with Q = softmax(z_g + log(P_p + 1e-6)).clamp_min(1e-8), `.mean()` per group, averaged
    over the groups that have children.  One fused HIP pass forward (hrseg_group_kl), one for the gradient w.r.t. the
    logits (hrseg_group_kl_bwd; the parent probabilities get none: the log-bias is constant inside a group).
    groups: [(parent_name, [child names])] of this level; levels_prev: parent names in channel order.  Opt-in:
    train.get_loss adds `lambda_kl` times it per level only when asked to."""
    if z_children_all is None or probs_prev_level is None or groups is None:
        return z_children_all.sum() * 0
    live = [(levels_prev.index(p), len(ch)) for p, ch in groups if len(ch) > 0]
    if not live:
        return z_children_all.sum() * 0
    if sum(g for _, g in live) != z_children_all.shape[1]:
        raise ValueError("group sizes do not add up to the level's channels")
    return _GroupKL.apply(z_children_all, probs_prev_level, [p for p, _ in live], [g for _, g in live])
