"""The five drop-in metric classes (reference Metrics/performance_metrics.py:27-141) and train.get_metrics
(train.py:38-81) on the GPU against oracle/metrics.py, on SEVERAL batches in a row: tensors of one step are freed
before the next one is made, so the caching allocator hands out the same device addresses again -- a result
cached by address would return the previous batch's counts."""
import argparse

import numpy as np
import pytest
import torch

from oracle import metrics as OM

pytestmark = pytest.mark.gpu


def _batch(seed, B=2, C0=4, C1=4, S=24):
    g = np.random.Generator(np.random.PCG64(seed))
    lab = g.integers(0, C0 - 1 + C1, (B, S, S))
    t0 = np.stack([lab == c for c in range(C0 - 1)] + [lab >= C0 - 1], 1).astype(np.float32)
    t1 = np.where((lab < C0 - 1)[:, None], -1.0,
                  np.stack([lab == C0 - 1 + c for c in range(C1)], 1)).astype(np.float32)
    p0 = g.random((B, C0, S, S)).astype(np.float32)
    p1 = g.random((B, C1, S, S)).astype(np.float32)
    return [p0, p1], [t0, t1]


def _metric_objects():
    from hrseg_amd.Metrics import performance_metrics as PP
    return PP.Accuracy(), PP.Jaccardindex(), PP.DiceScore(), PP.Precision(), PP.Recall()


def test_metric_classes_on_consecutive_batches_match_the_oracle():
    from hrseg_amd import train as PT
    acc, iou, dice, prec, rec = _metric_objects()
    args = argparse.Namespace(num_classes=[4, 4])
    dev = torch.device("cuda")
    seen = []
    for seed in (11, 12, 13):
        probs, targets = _batch(seed)
        want = {k: np.concatenate([OM.level_metrics(p, t, child_classes=(L > 0))[k]
                                   for L, (p, t) in enumerate(zip(probs, targets))]) for k in OM.METRIC_NAMES}
        pd = [torch.from_numpy(p).to(dev) for p in probs]
        td = [torch.from_numpy(t).to(dev) for t in targets]
        a_l, i_l, d_l, p_l, r_l = [], [], [], [], []
        cls = PT._new_class_metrics(8)
        _, _, _, _, _, _, no_bg = PT.get_metrics(pd, td, a_l, i_l, d_l, p_l, r_l, acc, iou, dice, prec, rec, dev, cls, args)
        for c in range(8):
            for k in OM.METRIC_NAMES:
                assert abs(cls[c][k][-1] - float(want[k][c])) < 1e-6, (seed, c, k)
        assert np.allclose(no_bg.cpu().numpy(), want["dice"][1:], atol=1e-6)
        assert abs(i_l[-1] - float(want["iou"].mean())) < 1e-6
        # the five objects called directly, outside get_metrics (a reference-shaped loop may do that)
        for L in (0, 1):
            for obj, k in ((acc, "accuracy"), (iou, "iou"), (dice, "dice"), (prec, "precision"), (rec, "recall")):
                got = obj(pd[L], td[L], dev, 4, L > 0).cpu().numpy()
                assert np.allclose(got, OM.level_metrics(probs[L], targets[L], L > 0)[k], atol=1e-6), (seed, L, k)
        seen.append((pd[0].data_ptr(), want["iou"].copy()))
        del pd, td                                     # freed: the next batch reuses the addresses
    assert any(not np.allclose(seen[0][1], s[1]) for s in seen[1:])     # the batches really differ
    assert len({s[0] for s in seen}) < len(seen), "allocator did not recycle an address: the scenario under test did not occur"


def test_in_place_refill_of_the_same_tensor_is_seen():
    """a loop that refills static buffers (graph replay, pinned staging) changes the data, not the tensor"""
    acc, iou, dice, prec, rec = _metric_objects()
    dev = torch.device("cuda")
    (p0, _), (t0, _) = _batch(21)
    pd, td = torch.from_numpy(p0).to(dev), torch.from_numpy(t0).to(dev)
    first = iou(pd, td, dev, 4, False).cpu().numpy()
    (q0, _), (u0, _) = _batch(22)
    pd.copy_(torch.from_numpy(q0))
    td.copy_(torch.from_numpy(u0))
    second = iou(pd, td, dev, 4, False).cpu().numpy()
    assert np.allclose(first, OM.level_metrics(p0, t0, False)["iou"], atol=1e-6)
    assert np.allclose(second, OM.level_metrics(q0, u0, False)["iou"], atol=1e-6)
    assert not np.allclose(first, second)


def test_ignore_index_edge_cases_known_answers():
    """hand-computed: (a) child level, a KEPT pixel predicted as background (no positive channel) counts as a miss
    for its class and as nothing else; (b) a class absent from targets and predictions scores 0, not NaN;
    (c) pixels whose target is the ignore label never count, whatever is predicted there."""
    acc, iou, dice, prec, rec = _metric_objects()
    dev = torch.device("cuda")
    # 3 child classes, 1x6 pixels. targets: c0 c0 c1 ign ign c1 ; predictions: c0 bg c1 c0 c2 c0
    t = np.zeros((1, 3, 1, 6), np.float32)
    p = np.zeros((1, 3, 1, 6), np.float32)
    for i, c in enumerate([0, 0, 1, None, None, 1]):
        if c is not None:
            t[0, c, 0, i] = 1
    for i, c in enumerate([0, None, 1, 0, 2, 0]):
        if c is not None:
            p[0, c, 0, i] = 1
    td, pd = torch.from_numpy(t).to(dev), torch.from_numpy(p).to(dev)
    # kept pixels 0,1,2,5: (c0,c0) (c0,bg) (c1,c1) (c1,c0) -> TP=[1,1,0] FP=[1,0,0] FN=[1,1,0]
    want = {"iou": [1 / 3, 1 / 2, 0], "dice": [1 / 2, 2 / 3, 0], "precision": [1 / 2, 1, 0], "recall": [1 / 2, 1 / 2, 0]}
    for obj, k in ((iou, "iou"), (dice, "dice"), (prec, "precision"), (rec, "recall"), (acc, "recall")):
        got = obj(pd, td, dev, 3, True).cpu().numpy()
        assert np.allclose(got, want[k], atol=1e-6), (k, got)
        assert np.allclose(OM.level_metrics(p, t, True)[k], want[k], atol=1e-6), k
    # the train loop's form of the same case: ignored pixels carry -1 in every channel and are zeroed first
    t2 = t.copy()
    t2[0, :, 0, 3:5] = -1
    from hrseg_amd import ops
    z = torch.from_numpy(np.where(p > 0, 5.0, -5.0).astype(np.float32)).to(dev)
    z[0, :, 0, 1] = torch.tensor([-5.0, -5.0, -5.0])            # all-equal logits: argmax 0 -> predicted c0 there
    _, cm = ops.predict_metrics(z, torch.from_numpy(t2).to(dev), child=True, mask_pred=True)
    cm = cm.cpu().numpy()
    assert cm[0].sum() == 2 and cm[0, 0] == 2       # ignored pixels: target background, prediction zeroed -> background
    assert cm.sum() == 6
