"""ORACLE (test infrastructure, not product): one full CPU training step.

Restates the per-batch body of /root/reference/train.py:179-246 on the oracle
models/losses/metrics (the committed call `get_loss(..., lambda_cons=...,
lambda_kl=...)` raises TypeError in the reference -- SURVEY.md D3 -- so those
kwargs are dropped; in training the consistency term sees the one-hot
predictions, D4).  Used by tests as the checker and by bench.py's
``cpu_baseline`` leg as the timed CPU port.
"""
from __future__ import annotations

import numpy as np
import torch

from . import losses as L
from . import metrics as M


def split_levels(target, num_classes):
    out, s = [], 0
    for n in num_classes:
        out.append(target[:, s:s + n])
        s += n
    return out


def forward_loss(model, x, target, num_classes, level_weights, hierarchical=True, is_unet=False,
                 with_metrics=True):
    """-> dict(loss, parts, cons, logits, probs, metrics)."""
    targets = split_levels(target, num_classes) if hierarchical else [target]
    if is_unet:
        probs, logits = model(x, type=1 if hierarchical else 0)
    else:
        probs, logits = model(x)
    if not hierarchical:
        logits = [logits]
    onehots = [torch.where(t == -1, torch.zeros_like(t),
                           torch.from_numpy(M.one_hot_predictions(z.detach().numpy())))
               for z, t in zip(logits, targets)]
    metrics = None
    if with_metrics:
        metrics = M.train_step_metrics([z.detach().numpy() for z in logits], [t.numpy() for t in targets])
    levels = getattr(model, "levels", None) if hierarchical else None
    parent_of = getattr(model, "parent_of", None) if hierarchical else None
    loss, parts, cons = L.get_loss(logits, targets, level_weights,
                                   probs_per_level=onehots if hierarchical else None,
                                   levels=levels, parent_of=parent_of)
    return dict(loss=loss, parts=parts, cons=cons, logits=logits, probs=probs, metrics=metrics)


def train_step(model, optimizer, x, target, num_classes, level_weights, hierarchical=True, is_unet=False,
               with_metrics=True):
    model.train()
    optimizer.zero_grad()
    out = forward_loss(model, x, target, num_classes, level_weights, hierarchical, is_unet, with_metrics)
    out["loss"].backward()
    optimizer.step()
    return out
