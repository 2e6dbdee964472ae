"""Hand-computed known-answer cases for the metric definitions (the torchmetrics boundary of the
reference is unpinned -- see oracle/metrics.py): both the numpy oracle and the product's
confusion-matrix reduction must reproduce them.  CPU only."""
import numpy as np
import torch

from oracle import metrics as OM


def onehot(labels, c):
    return np.moveaxis(np.eye(c, dtype=np.float32)[labels], -1, 1)


def test_level0_known_answer():
    tgt = np.array([[[0, 1], [2, 1]]])            # [B=1,2,2]
    prd = np.array([[[0, 2], [2, 1]]])
    m = OM.level_metrics(onehot(prd, 3), onehot(tgt, 3), child_classes=False)
    # TP=[1,1,1] FP=[0,0,1] FN=[0,1,0]
    assert np.allclose(m["iou"], [1, 0.5, 0.5])
    assert np.allclose(m["dice"], [1, 2 / 3, 2 / 3])
    assert np.allclose(m["precision"], [1, 1, 0.5])
    assert np.allclose(m["recall"], [1, 0.5, 1])
    assert np.allclose(m["accuracy"], m["recall"])


def test_child_level_ignores_background_targets_and_counts_masked_predictions():
    # 2 child classes; pixels: tgt = [c0, c1, ignored(-1,-1), c0]; pred one-hot = [c0, c0, c1, none(all zero)]
    t = np.array([[[1, 0, -1, 1]], [[0, 1, -1, 0]]], dtype=np.float32).reshape(1, 2, 1, 4)
    p = np.array([[[1, 1, 0, 0]], [[0, 0, 1, 0]]], dtype=np.float32).reshape(1, 2, 1, 4)
    t_eval = np.where(t == -1, 0, t)
    p_eval = np.where(t == -1, 0, p)              # train loop zeroes predictions where target == -1
    m = OM.level_metrics(p_eval, t_eval, child_classes=True)
    # kept pixels (target != background): #0 (c0,c0) #1 (c1,c0) #3 (c0,bg)  -> TP=[1,0] FP=[1,0] FN=[1,1]
    assert np.allclose(m["iou"], [1 / 3, 0])
    assert np.allclose(m["dice"], [0.5, 0])
    assert np.allclose(m["precision"], [0.5, 0])
    assert np.allclose(m["recall"], [0.5, 0])


def test_empty_class_and_all_ignored_are_zero_not_nan():
    t = np.full((1, 3, 2, 2), -1.0, np.float32)
    p = onehot(np.zeros((1, 2, 2), int), 3)
    m = OM.level_metrics(np.where(t == -1, 0, p), np.where(t == -1, 0, t), child_classes=True)
    for k in OM.METRIC_NAMES:
        assert np.array_equal(m[k], np.zeros(3, np.float32)), k


def test_product_confusion_reduction_matches_the_known_answers():
    from hrseg_amd.Metrics.performance_metrics import metrics_from_confusion
    cm0 = torch.tensor([[1, 0, 0], [0, 1, 1], [0, 0, 1]])          # cm[target, pred] of the level-0 case
    m = metrics_from_confusion(cm0, False)
    assert np.allclose(m["iou"].numpy(), [1, 0.5, 0.5]) and np.allclose(m["dice"].numpy(), [1, 2 / 3, 2 / 3])
    assert np.allclose(m["precision"].numpy(), [1, 1, 0.5]) and np.allclose(m["recall"].numpy(), [1, 0.5, 1])
    # child case: labels shifted by one, row 0 = ignored targets (one pixel predicted c1 -> label 2)
    cm1 = torch.tensor([[0, 0, 1], [1, 1, 0], [0, 1, 0]])
    m = metrics_from_confusion(cm1, True)
    assert np.allclose(m["iou"].numpy(), [1 / 3, 0]) and np.allclose(m["dice"].numpy(), [0.5, 0])
    assert np.allclose(m["precision"].numpy(), [0.5, 0]) and np.allclose(m["recall"].numpy(), [0.5, 0])
    z = metrics_from_confusion(torch.zeros(4, 4, dtype=torch.int64), True)
    assert all(float(v.abs().sum()) == 0 for v in z.values())


def test_train_step_metrics_matches_manual_pipeline():
    g = np.random.Generator(np.random.PCG64(3))
    z0, z1 = g.standard_normal((2, 4, 9, 9)).astype(np.float32), g.standard_normal((2, 4, 9, 9)).astype(np.float32)
    lab = g.integers(0, 7, (2, 9, 9))
    t0 = np.stack([lab == 0, lab == 1, lab == 2, lab >= 3], 1).astype(np.float32)
    t1 = np.where((lab < 3)[:, None], -1.0, np.stack([lab == 3, lab == 4, lab == 5, lab == 6], 1)).astype(np.float32)
    m = OM.train_step_metrics([z0, z1], [t0, t1])
    # recompute level-1 recall of class 0 by brute force
    pred1 = z1.argmax(1)
    keep = lab >= 3
    tp = np.sum((lab == 3) & (pred1 == 0) & keep)
    fn = np.sum((lab == 3) & (pred1 != 0) & keep)
    assert abs(m["recall"][4] - tp / max(tp + fn, 1)) < 1e-6
