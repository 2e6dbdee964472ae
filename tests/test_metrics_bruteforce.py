"""A SECOND, independent restatement of the reference's per-class metrics (Metrics/performance_metrics.py:27-141): a per-pixel
Python loop over the documented torchmetrics semantics (task='multiclass', average=None, ignore_index) -- no confusion matrix,
no vectorised arithmetic, nothing shared with oracle/metrics.py or the kernels -- property-tested against the oracle (CPU) and
against hrseg_predict_metrics + hrseg_metric_vectors and the metric classes (GPU) on random shapes, including all-ignored
inputs, empty classes and child levels whose pixels lie outside the parent.

PARITY STAYS UNPINNED at the torchmetrics boundary: torchmetrics is not installed, not vendored and not pinned by the
reference, and the reference holds no fixtures; two independent restatements agreeing with each other and with hand-computed
known answers (tests/test_metrics_known_answers.py) is as far as this environment allows."""
import numpy as np
import pytest

NAMES = ("accuracy", "iou", "dice", "precision", "recall")


def brute_force_level_metrics(probs, targets, child_classes):
    """per-pixel loop.  ProcessClasses (:27-47): arg-max label of prediction and target (first maximum); child levels get a
    synthetic class 0 = 'no positive channel'.  Then per class c over the pixels whose TARGET label is not ignore_index
    (0 for child levels -- and class 0 itself is dropped from the report --, -1 i.e. none for the root level):
    tp = pred == c and target == c, fp = pred == c and target != c, fn = pred != c and target == c;
    accuracy = recall = tp / (tp + fn), precision = tp / (tp + fp), iou = tp / (tp + fp + fn), dice = 2 tp / (2 tp + fp + fn),
    0 when the denominator is 0."""
    B, C, H, W = probs.shape
    K = C + 1 if child_classes else C
    tp, fp, fn = [0] * K, [0] * K, [0] * K
    for b in range(B):
        for y in range(H):
            for x in range(W):
                pv = [float(probs[b, c, y, x]) for c in range(C)]
                tv = [float(targets[b, c, y, x]) for c in range(C)]
                if child_classes:
                    pv = [1.0 if sum(pv) == 0 else 0.0] + pv
                    tv = [1.0 if sum(tv) == 0 else 0.0] + tv
                pl = max(range(K), key=lambda i: (pv[i], -i))
                tl = max(range(K), key=lambda i: (tv[i], -i))
                if child_classes and tl == 0:
                    continue                                  # ignore_index = 0
                for c in range(K):
                    if pl == c and tl == c:
                        tp[c] += 1
                    elif pl == c:
                        fp[c] += 1
                    elif tl == c:
                        fn[c] += 1
    lo = 1 if child_classes else 0

    def div(a, b):
        return np.float32(a / b) if b else np.float32(0.0)
    out = {k: [] for k in NAMES}
    for c in range(lo, K):
        out["accuracy"].append(div(tp[c], tp[c] + fn[c]))
        out["recall"].append(div(tp[c], tp[c] + fn[c]))
        out["precision"].append(div(tp[c], tp[c] + fp[c]))
        out["iou"].append(div(tp[c], tp[c] + fp[c] + fn[c]))
        out["dice"].append(div(2 * tp[c], 2 * tp[c] + fp[c] + fn[c]))
    return {k: np.array(v, dtype=np.float32) for k, v in out.items()}


def _cases():
    """(tag, one-hot-ish predictions, targets {0,1}, child flag) -- as the train loop hands them over (train.py:206-232):
    predictions and targets already zeroed where the raw target was -1"""
    g = np.random.Generator(np.random.PCG64(2024))
    out = []
    for i, (B, C, H, W, child) in enumerate([(2, 4, 9, 7, False), (1, 4, 12, 5, True), (3, 3, 6, 6, True), (2, 7, 8, 8, False),
                                             (1, 2, 5, 11, True), (2, 5, 7, 9, False)]):
        lab_p = g.integers(0, C, size=(B, H, W))
        lab_t = g.integers(0, C, size=(B, H, W))
        p = np.moveaxis(np.eye(C, dtype=np.float32)[lab_p], -1, 1).copy()
        t = np.moveaxis(np.eye(C, dtype=np.float32)[lab_t], -1, 1).copy()
        if child:                              # pixels outside the parent: target -1 -> both zeroed by the prediction prep
            outside = g.random((B, H, W)) < 0.35
            p[np.broadcast_to(outside[:, None], p.shape)] = 0.0
            t[np.broadcast_to(outside[:, None], t.shape)] = 0.0
        out.append((f"random{i}", p, t, child))
    # a class that never occurs in the target and one that is never predicted
    p, t = out[0][1].copy(), out[0][2].copy()
    t[:, 2] = 0.0
    t[:, 0] = np.maximum(t[:, 0], 1.0 - t.sum(1))
    p[:, 1] = 0.0
    p[:, 3] = np.maximum(p[:, 3], 1.0 - p.sum(1))
    out.append(("empty_classes", p, t, False))
    # every pixel ignored (child level, nothing inside the parent)
    out.append(("all_ignored_child", np.zeros((2, 4, 6, 6), np.float32), np.zeros((2, 4, 6, 6), np.float32), True))
    # predictions inside, targets all background: only false positives on dropped pixels -> all zeros
    p = out[1][1].copy()
    out.append(("target_all_background", p, np.zeros_like(p), True))
    # soft probabilities (the test() loop hands over probabilities, train.py:337-340): arg-max with ties -> first
    ps = g.random((2, 4, 7, 7)).astype(np.float32)
    ps[0, :, 0, 0] = 0.25
    out.append(("soft_probs_root", ps, out[0][2][:, :, :7, :7].copy(), False))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_oracle_metrics_equal_the_brute_force_restatement(case):
    from oracle import metrics as OM
    _, p, t, child = case
    want = brute_force_level_metrics(p, t, child)
    got = OM.level_metrics(p, t, child_classes=child)
    for k in NAMES:
        assert np.array_equal(got[k], want[k]), (k, got[k], want[k])


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_gpu_metrics_equal_the_brute_force_restatement(case):
    """the metric classes (performance_metrics.py call surface) and the fused path of the train loop
    (hrseg_predict_metrics counts -> hrseg_metric_vectors) against the per-pixel loop"""
    import torch
    from hrseg_amd import ops, train as PT
    from hrseg_amd.Metrics import performance_metrics as PP
    tag, p, t, child = case
    want = brute_force_level_metrics(p, t, child)
    pd, td = torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda()
    C = p.shape[1]
    fns = {"accuracy": PP.Accuracy(), "iou": PP.Jaccardindex(), "dice": PP.DiceScore(), "precision": PP.Precision(),
           "recall": PP.Recall()}
    for k in NAMES:
        got = fns[k](pd, td, torch.device("cuda"), C, child).cpu().numpy()
        assert got.shape == (C,) and np.allclose(got, want[k], atol=1e-7), (tag, k, got, want[k])
    # the counting kernel in its test()-loop form (raw probabilities / targets, no masking) + the vector kernel
    _, cm = ops.predict_metrics(pd, td, child=child, mask_pred=False, want_onehot=False)
    vec = PT._metric_vectors([cm] if not child else [torch.zeros((C, C), dtype=torch.int64, device="cuda"), cm])
    for i, k in enumerate(PP.METRIC_NAMES):
        got = vec[k].cpu().numpy()
        got = got[C:] if child else got
        assert np.allclose(got, want[k], atol=1e-7), (tag, k, got, want[k])


@pytest.mark.gpu
def test_gpu_train_loop_metrics_equal_the_brute_force_restatement_on_logits():
    """train loop form: logits + ternary targets -> masked one-hot + counts in ONE kernel (mask_pred=1); the brute force sees
    what the reference's prediction prep would hand to the metric classes (train.py:206-231)"""
    import torch
    from hrseg_amd import ops, train as PT
    from hrseg_amd.Metrics import performance_metrics as PP
    g = np.random.Generator(np.random.PCG64(99))
    for child in (False, True):
        B, C, H, W = 2, 4, 10, 9
        z = g.standard_normal((B, C, H, W)).astype(np.float32)
        lab = g.integers(0, C, size=(B, H, W))
        t = np.moveaxis(np.eye(C, dtype=np.float32)[lab], -1, 1).copy()
        if child:
            t[np.broadcast_to((g.random((B, H, W)) < 0.4)[:, None], t.shape)] = -1.0
        else:
            t[0, 1, :3] = -1.0                                  # a few ignored entries on the root level too
        oh = np.moveaxis(np.eye(C, dtype=np.float32)[z.argmax(1)], -1, 1)
        p_in = np.where(t == -1, 0.0, oh).astype(np.float32)
        t_in = np.where(t == -1, 0.0, t).astype(np.float32)
        want = brute_force_level_metrics(p_in, t_in, child)
        onehot, cm = ops.predict_metrics(torch.from_numpy(z).cuda(), torch.from_numpy(t).cuda(), child=child, mask_pred=True)
        assert np.array_equal(onehot.cpu().numpy(), p_in)
        vec = PT._metric_vectors([cm] if not child else [torch.zeros((C, C), dtype=torch.int64, device="cuda"), cm])
        for k in PP.METRIC_NAMES:
            got = vec[k].cpu().numpy()
            got = got[C:] if child else got
            assert np.allclose(got, want[k], atol=1e-7), (child, k, got, want[k])
