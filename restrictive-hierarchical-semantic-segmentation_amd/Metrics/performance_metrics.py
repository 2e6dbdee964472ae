"""Per-step segmentation metrics on the GPU.

Same call surface as the reference's Metrics/performance_metrics.py:27-141
(``Accuracy, Jaccardindex, DiceScore, Precision, Recall`` called as
``m(probs, targets, device, num_classes, child_classes=False) -> [num_classes]``),
but one HIP pass (hrseg_predict_metrics) builds the (target, prediction)
confusion counts that all five share, instead of five argmax passes and five
torchmetrics objects per level per step.

Definitions (torchmetrics multiclass, average=None, as the reference uses it):
pixels whose target label equals ignore_index are dropped (child levels:
the synthetic background label 0, then entry 0 is sliced off; level 0: -1,
i.e. nothing); IoU=TP/(TP+FP+FN), Dice=F1=2TP/(2TP+FP+FN), Prec=TP/(TP+FP),
Rec=Acc=TP/(TP+FN), 0 where a denominator is 0.
"""
from __future__ import annotations

import torch

from .. import ops

METRIC_NAMES = ("accuracy", "iou", "dice", "precision", "recall")


def _safe_div(a, b):
    return torch.where(b == 0, torch.zeros_like(a), a / torch.where(b == 0, torch.ones_like(b), b))


def metrics_from_confusion(cm: torch.Tensor, child_classes: bool):
    """cm[target, pred] int64 [K,K] -> dict name -> [C] fp32 (device tensor, no sync)."""
    cm = cm.to(torch.float64)
    if child_classes:
        cm = cm[1:]                       # drop pixels whose target is the background label 0
        tp = torch.diagonal(cm[:, 1:])
        fn = cm.sum(1) - tp
        fp = cm.sum(0)[1:] - tp
    else:
        tp = torch.diagonal(cm)
        fn = cm.sum(1) - tp
        fp = cm.sum(0) - tp
    rec = _safe_div(tp, tp + fn).float()
    return {"accuracy": rec, "iou": _safe_div(tp, tp + fp + fn).float(),
            "dice": _safe_div(2 * tp, 2 * tp + fp + fn).float(),
            "precision": _safe_div(tp, tp + fp).float(), "recall": rec}


# The five metric objects are called back to back on the same (probs, targets) pair (reference
# train.py:47-51).  Inside `shared_confusion()` -- entered by train.get_metrics around exactly those five
# calls -- the counts of a pair are computed once and shared; the scope holds the tensors themselves, so a
# key can never outlive its tensors (device addresses are recycled by the caching allocator from one step to
# the next and the kernels write through raw pointers, so neither data_ptr nor _version identifies a batch).
# Outside a scope every call counts afresh.
_scope = None


class shared_confusion:
    """with shared_confusion(): ...  -- metric calls on the same tensor objects share one counting pass"""

    def __enter__(self):
        global _scope
        self._outer, _scope = _scope, {}
        return self

    def __exit__(self, *exc):
        global _scope
        _scope = self._outer
        return False


def confusion_for(probs, targets, child_classes):
    """Confusion counts [K,K] of (argmax targets, argmax probs), K = C + child"""
    key = (id(probs), id(targets), bool(child_classes))
    if _scope is not None and key in _scope:
        return _scope[key][0]
    p = probs.contiguous().float()
    t = targets.contiguous().float()
    _, cm = ops.predict_metrics(p, t, child=bool(child_classes), mask_pred=False, want_onehot=False)
    if _scope is not None:
        _scope[key] = (cm, probs, targets)          # the references keep the ids unique within the scope
    return cm


class _Metric(torch.nn.Module):
    name = ""

    def __init__(self, smooth=1):
        super().__init__()

    def forward(self, probs, targets, device, num_classes, child_classes=False):
        cm = confusion_for(probs, targets, child_classes)
        return metrics_from_confusion(cm, child_classes)[self.name]


class DiceScore(_Metric):
    name = "dice"


class Jaccardindex(_Metric):
    name = "iou"


class Accuracy(_Metric):
    name = "accuracy"


class Precision(_Metric):
    name = "precision"


class Recall(_Metric):
    name = "recall"
