"""ORACLE (test infrastructure, not product): CPU restatement of the reference losses.

Closed forms of /root/reference/Metrics/losses.py:
  CrossEntropyLoss              :90-134  (masked, class-weighted, NaN item -> 1.0)
  SoftDiceLoss                  :16-86   (smooth 0, NaN items dropped, may be None)
  hierarchical_consistency_loss :150-177
and get_loss of /root/reference/train.py:111-152.

Parity: pinned by tests/golden/*.npz generated from the imported reference.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

IGNORE = -1.0


def _flat(x):
    return x.contiguous().view(x.size(0), x.size(1), -1)


def _weights(class_weight, like):
    return torch.tensor(class_weight, dtype=torch.float32, device=like.device)


def cross_entropy_loss(outs, targets, logits_input=False, class_weight=None):
    """mean_b [ (1/C) sum_c  -(w_c / n_bc) sum_{t!=-1} t*logp ]  with NaN items -> 1."""
    logp = _flat(F.log_softmax(outs, dim=1) if logits_input else outs)
    t = _flat(targets)
    m = (t != IGNORE).to(logp.dtype)
    w = _weights(class_weight, logp)                        # [C]
    n = m.sum(-1)                                           # [B,C]
    s = (t * logp * m).sum(-1)
    # an empty (b,c) mask makes the reference's masked mean NaN, the item is then
    # replaced by the constant 1.0 and carries no gradient (losses.py:116)
    valid = (n > 0).all(1)
    item = (-(w[None, :] * s) / torch.where(n > 0, n, torch.ones_like(n))).sum(1) / logp.shape[1]
    item = torch.where(valid, item, torch.ones_like(item))
    return item.mean()


def soft_dice_loss(outs, targets, logits_input=False, class_weight=None):
    """mean over non-NaN b of 1 - 2 I_b / U_b; None when every item is NaN."""
    p = _flat(F.softmax(outs, dim=1) if logits_input else outs)
    t = _flat(targets)
    m = (t != IGNORE).to(p.dtype)
    w = _weights(class_weight, p)
    inter = (w[None, :] * (p * t * m).sum(-1)).sum(1)
    union = (w[None, :] * ((p * m).sum(-1) + (t * m).sum(-1))).sum(1)
    keep = union != 0                                        # 0/0 items are dropped (losses.py:64)
    if not bool(keep.any()):
        return None
    item = 1.0 - 2.0 * inter / torch.where(keep, union, torch.ones_like(union))
    return item[keep].mean()


def hierarchical_consistency_loss(probs_per_level, levels, parent_of, reduction="mean"):
    if probs_per_level is None or levels is None or parent_of is None:
        return probs_per_level[0].sum() * 0 if probs_per_level else 0.0
    total, count = 0.0, 0
    for L in range(1, len(levels)):
        prev, cur = probs_per_level[L - 1], probs_per_level[L]
        for p_idx, p_name in enumerate(levels[L - 1]):
            idx = [i for i, c in enumerate(levels[L]) if parent_of.get(c) == p_name]
            if not idx:
                continue
            diff = (cur[:, idx].sum(1, keepdim=True) - prev[:, p_idx:p_idx + 1]).abs()
            total = total + (diff.mean() if reduction == "mean" else diff.sum())
            count += 1
    if count == 0:
        return probs_per_level[0].sum() * 0
    return total / count


def get_loss(output_logits, targets, level_weights, probs_per_level=None, levels=None, parent_of=None,
             cur_epoch=None, pretrain_epoch=None):
    """-> (loss, per-level [ce, dice], consistency).  train.py:111-152."""
    n = len(output_logits)
    cap = n - 1 if pretrain_epoch is None else int(min(n - 1, cur_epoch // pretrain_epoch))
    loss, parts = 0.0, []
    for L in range(n):
        if L > cap:
            parts.append((None, None))
            continue
        ce = cross_entropy_loss(output_logits[L], targets[L], True, level_weights[L])
        dice = soft_dice_loss(output_logits[L], targets[L], True, level_weights[L])
        loss = loss + ce
        if dice is not None:
            loss = loss + dice
        parts.append((ce, dice))
    cons = None
    if probs_per_level is not None and levels is not None and parent_of is not None:
        cons = hierarchical_consistency_loss(probs_per_level, levels, parent_of)
        loss = loss + cons
    return loss, parts, cons


def grouped_conditional_kl(z_children_all, probs_prev_level, groups, levels_prev):
    """restates the commented-out stabiliser of the reference (Metrics/losses.py:180-210) with stock torch ops:
    per parent group KL(softmax(z_g + log(P_p + 1e-6)).clamp_min(1e-8) || Uniform).mean(), averaged over groups"""
    if z_children_all is None or probs_prev_level is None or groups is None:
        return z_children_all.sum() * 0
    kl, gcount, start = 0.0, 0, 0
    for pname, chnames in groups:
        g = len(chnames)
        if g == 0:
            continue
        z_g = z_children_all[:, start:start + g]
        p_p = probs_prev_level[:, levels_prev.index(pname):levels_prev.index(pname) + 1]
        q = torch.softmax(z_g + torch.log(p_p + 1e-6), dim=1).clamp_min(1e-8)
        u = torch.full_like(q, 1.0 / g)
        kl = kl + (q * (q.log() - u.log())).mean()
        gcount += 1
        start += g
    if gcount == 0:
        return z_children_all.sum() * 0
    return kl / gcount
