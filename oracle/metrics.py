"""ORACLE (test infrastructure, not product): per-step segmentation metrics on CPU (numpy).

Restates /root/reference/Metrics/performance_metrics.py:27-141 and the
prediction prep + get_metrics of /root/reference/train.py:38-81, 206-231.

PARITY UNPINNED at the torchmetrics boundary: the reference delegates the
arithmetic to torchmetrics (third-party, unpinned in requirements.txt, not
installed here, absent from /root/reference) and holds no tests.  This file
restates the documented multiclass definitions -- drop pixels whose *target*
label equals ignore_index, global per-class TP/FP/FN over the batch,
IoU=TP/(TP+FP+FN), F1=2TP/(2TP+FP+FN), Prec=TP/(TP+FP), Rec=Acc=TP/(TP+FN),
0 when a denominator is 0 -- and tests pin it with hand-computed
confusion-matrix cases (tests/test_metrics_known_answers.py).
"""
from __future__ import annotations

import numpy as np

METRIC_NAMES = ("accuracy", "iou", "dice", "precision", "recall")


def process_classes(probs: np.ndarray, targets: np.ndarray, child_classes: bool):
    """[B,C,H,W] x2 -> label maps; child levels get a synthetic class 0 where
    a map has no positive channel (performance_metrics.py:31-47)."""
    if child_classes:
        pb = (probs.sum(1, keepdims=True) == 0).astype(probs.dtype)
        tb = (targets.sum(1, keepdims=True) == 0).astype(targets.dtype)
        probs = np.concatenate([pb, probs], 1)
        targets = np.concatenate([tb, targets], 1)
    return probs.argmax(1), targets.argmax(1)


def confusion_counts(pred: np.ndarray, tgt: np.ndarray, n_classes: int, ignore_index: int):
    keep = tgt != ignore_index
    pred, tgt = pred[keep].ravel(), tgt[keep].ravel()
    cm = np.bincount(tgt * n_classes + pred, minlength=n_classes * n_classes).reshape(n_classes, n_classes)
    tp = np.diag(cm).astype(np.int64)
    fp = cm.sum(0) - tp
    fn = cm.sum(1) - tp
    return tp, fp, fn


def _div(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.where(b == 0, 0.0, a / np.where(b == 0, 1.0, b)).astype(np.float32)


def level_metrics(probs: np.ndarray, targets: np.ndarray, child_classes: bool):
    """-> dict name -> [C] fp32 for one level."""
    c = targets.shape[1]
    pred, tgt = process_classes(probs, targets, child_classes)
    if child_classes:
        tp, fp, fn = confusion_counts(pred, tgt, c + 1, 0)
        tp, fp, fn = tp[1:], fp[1:], fn[1:]
    else:
        tp, fp, fn = confusion_counts(pred, tgt, c, -1)
    rec = _div(tp, tp + fn)
    return {"accuracy": rec, "iou": _div(tp, tp + fp + fn), "dice": _div(2 * tp, 2 * tp + fp + fn),
            "precision": _div(tp, tp + fp), "recall": rec}


def one_hot_predictions(logits: np.ndarray) -> np.ndarray:
    """softmax -> argmax -> one-hot NCHW float (train.py:206-224); softmax is
    monotone so the argmax is taken on the logits."""
    idx = logits.argmax(1)
    return np.moveaxis(np.eye(logits.shape[1], dtype=np.float32)[idx], -1, 1)


def train_step_metrics(logits_per_level, targets_per_level):
    """Prediction prep + get_metrics for the train loop (train.py:206-232):
    predictions and targets are zeroed where target == -1."""
    per_metric = {k: [] for k in METRIC_NAMES}
    for L, (z, t) in enumerate(zip(logits_per_level, targets_per_level)):
        pred = np.where(t == -1, 0.0, one_hot_predictions(z)).astype(np.float32)
        tgt = np.where(t == -1, 0.0, t).astype(np.float32)
        m = level_metrics(pred, tgt, child_classes=(L > 0))
        for k in METRIC_NAMES:
            per_metric[k].append(m[k])
    return {k: np.concatenate(v) for k, v in per_metric.items()}
