"""Product models / losses / train step on the GPU against (a) the golden vectors generated from
the reference and (b) the CPU oracle on the same seeded inputs.  Bar: 1e-3 relative (fp32),
as BASELINE.json's north_star states."""
import argparse

import os

import numpy as np
import pytest
import torch

from tests.helpers import CASES, CONV_MODES, build_model, conv_mode, level_weights_for, load_golden, load_tree, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3
_ORACLE_STEPS = {}       # golden case -> (the CPU oracle's two train steps, its state_dict after them)
_ORACLE_RAGGED = {}


def _args(kind, hier, num_classes, weights, batch):
    return argparse.Namespace(model_type=1 if hier else 0, model_select=0 if kind == "unet" else 1,
                              num_classes=num_classes, level_weights=weights, level0_pretrain_epochs=None,
                              batch_size=batch)


@pytest.mark.parametrize("mode", CONV_MODES)
@pytest.mark.parametrize("name", list(CASES))
def test_model_matches_reference_golden(name, mode):
    """every golden case under every convolution arithmetic: the kernels that produce the headline number (mode
    "auto_ws": wave-specialised group launches, fp16x2 im2col kernels, nine-tap weight gradients) are pinned by the
    reference's own vectors, not only the exact-fp32 kernels the default thresholds route small cases to"""
    from hrseg_amd.Models import models as PM
    kind, hier, tree_file, size, batch = CASES[name]
    model = build_model(PM, kind, hier, load_tree(tree_file), size).cuda()
    with conv_mode(model, mode) as cm:
        _golden_body(name, model, mode)
    cm.check_families(kind)


def _golden_body(name, model, mode="auto"):
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    assert [n for n, _ in model.named_parameters()] == list(g["grad_names"])
    x, target = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()
    num_classes = [int(v) for v in g["num_classes"]]
    weights = level_weights_for(tree_file, hier)
    args = _args(kind, hier, num_classes, weights, batch)

    model.eval()
    with torch.no_grad():
        _, logits = PT._model_call(model, x, args, tree)
    logits = logits if hier else [logits]
    for L, z in enumerate(logits):
        assert rel_err(z.cpu().numpy(), g[f"eval_logits{L}"]) < TOL, f"eval logits {L}"

    model.train()
    probs, logits = PT._model_call(model, x, args, tree)
    logits = logits if hier else [logits]
    targets = PT.split_targets(target, args)
    loss = 0.0
    onehots = []
    for L, (z, t) in enumerate(zip(logits, targets)):
        assert rel_err(z.detach().cpu().numpy(), g[f"logits{L}"]) < TOL, f"logits {L}"
        res = PL.fused_ce_dice(z, t, weights[L])
        assert abs(float(res[0]) - g[f"ce{L}"]) < TOL * max(1.0, abs(g[f"ce{L}"]))
        assert abs(float(res[1]) - g[f"dice{L}"]) < TOL * max(1.0, abs(g[f"dice{L}"]))
        loss = loss + res[0] + res[1]
        from hrseg_amd import ops
        oh, _ = ops.predict_metrics(z.detach(), t, child=(L > 0), mask_pred=True)
        # arg-max ties/near-ties can flip a pixel: allow a 1e-3 fraction of disagreeing pixels
        assert float((oh.cpu() != torch.from_numpy(g[f"onehot{L}"])).float().mean()) < 1e-3
        onehots.append(oh)
    if hier:
        for L, p in enumerate(probs):
            assert rel_err(p.detach().cpu().numpy(), g[f"probs{L}"]) < TOL, f"probs {L}"
        cons = PL.hierarchical_consistency_loss(onehots, model.levels, model.parent_of)
        assert abs(float(cons) - g["cons_onehot"]) < 2e-3
        cons_p = PL.hierarchical_consistency_loss([p.detach() for p in probs], model.levels, model.parent_of)
        assert abs(float(cons_p) - g["cons_probs"]) < 1e-5
        loss = loss + cons            # the value just computed (train.get_loss adds exactly this term)
    assert abs(float(loss) - g["loss"]) < TOL * abs(g["loss"])

    loss.backward()
    named = dict(model.named_parameters())
    norms = np.array([0.0 if p.grad is None else float(p.grad.double().norm()) for p in named.values()])
    ref = g["grad_norms"]
    scale = np.maximum(ref, 1e-2 * ref.max())
    worst = np.argmax(np.abs(norms - ref) / scale)
    # outputs, losses and BN buffers above are held at 1e-3 in every mode.  Gradient NORMS of the earliest layers sit on the
    # fp32 noise floor of these 32..64-pixel nets (2 x 2-pixel lowest branch, BN over 8 samples; tests/
    # test_grad_noise_gpu.py asserts it: the CPU-fp32 reference is ~1e-2 from an fp64 evaluation there): 5e-3 for the default routing and the
    # exact-fp32 kernels, 1e-2 where every layer is forced onto the split-precision kernels with their split-K atomics
    # (observed: 7.4e-3 on stem.1.weight of hrnet_flat_tl_64 in one run of four)
    bar = 1e-2 if mode in ("fp16x2", "auto_ws") else 5e-3
    assert np.abs(norms - ref)[worst] / scale[worst] < bar, (list(named)[worst], norms[worst], ref[worst])
    for key in g.files:
        if key.startswith("grad::"):
            got = named[key[6:]].grad.cpu().numpy()
            # head / FiLM gradients are a few ops from the loss; the first conv's sits behind every
            # BN/ReLU of the net, where fp32 evaluations differ from each other by ~1e-2 already
            # (tests/test_grad_noise_gpu.py: CPU-fp32 and GPU are equally far from an fp64 evaluation)
            tol = 5e-2 if key[6:].startswith(("stem", "inc0")) else 5e-3
            # absolute floor 2e-6: a head bias gradient is a sum over all pixels that cancels almost completely (the two-class
            # head of hrnet_hier_ext_62: +-8.9e-6 from terms of ~1e-3), so its value carries the rounding of that sum --
            # observed 1.15e-6 from the golden in one run of the default routing, 0.3e-6 in others
            assert np.abs(got - g[key]).max() < tol * np.abs(g[key]).max() + 2e-6, key
    bufs = np.array([float(b.double().norm()) for _, b in model.named_buffers()])
    assert np.max(np.abs(bufs - g["buf_norms"]) / np.maximum(g["buf_norms"], 1e-6)) < TOL


@pytest.mark.parametrize("mode", CONV_MODES)
@pytest.mark.parametrize("name", ["unet_hier_tl_62", "hrnet_hier_tl_64"])
def test_train_steps_track_the_oracle(name, mode):
    """two full train steps (fwd, metrics, loss, bwd, AdamW) next to the CPU oracle, under every convolution arithmetic"""
    from oracle import models as OM
    from oracle import train_step as OT
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    num_classes = [int(v) for v in g["num_classes"]]
    weights = level_weights_for(tree_file, hier)
    args = _args(kind, hier, num_classes, weights, batch)
    x, target = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])

    if name not in _ORACLE_STEPS:          # the oracle's two steps do not depend on the product's convolution mode: run once
        om = build_model(OM, kind, hier, tree, size)
        oopt = torch.optim.AdamW(om.parameters(), lr=1e-4)
        refs = [OT.train_step(om, oopt, x, target, num_classes, weights, hierarchical=hier, is_unet=(kind == "unet"))
                for _ in range(2)]
        refs = [dict(loss=r["loss"].detach().clone(), metrics={k: np.array(v) for k, v in r["metrics"].items()}) for r in refs]
        # the same two steps in fp64: the yardstick for the BN running statistics after the second step (below)
        o64 = build_model(OM, kind, hier, tree, size).double()
        opt64 = torch.optim.AdamW(o64.parameters(), lr=1e-4)
        for _ in range(2):
            OT.train_step(o64, opt64, x.double(), target.double(), num_classes, weights, hierarchical=hier,
                          is_unet=(kind == "unet"), with_metrics=False)
        _ORACLE_STEPS[name] = (refs, {n: v.detach().clone() for n, v in om.state_dict().items()},
                               {n: v.detach().clone() for n, v in o64.state_dict().items() if "running_" in n})
    refs, osd, osd64 = _ORACLE_STEPS[name]
    pm = build_model(PM, kind, hier, tree, size).cuda()
    popt = PT.FusedAdamW(pm, lr=[1e-4])
    loss_fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in num_classes]
    pm.train()
    level_loss = []
    with conv_mode(pm, mode) as cm_ctx:
        for step in range(2):
            ref = refs[step]
            loss, cms = PT.train_step(pm, popt, x.cuda(), target.cuda(), loss_fns, args, tree, level_loss)
            assert abs(float(loss) - ref["loss"].item()) < TOL * abs(ref["loss"].item()), f"step {step}"
            vec = PT._metric_vectors(cms)
            for k, v in ref["metrics"].items():
                got = vec[k].cpu().numpy()
                if step == 0:
                    assert np.allclose(got, v, atol=2e-3), (step, k)
                else:
                    # after one AdamW step the two evaluations' weights differ by rounding noise (+-lr on elements whose
                    # gradient is noise, fp32 atomics in the default mode): a pixel at a decision boundary may flip, and
                    # one pixel of a 64 x 64 image moves a rare class's precision / recall by 1 / count
                    # (the tight second-step statement is test_train_steps_track_the_oracle_at_256)
                    assert np.allclose(got, v, atol=5e-2) and np.abs(got - v).mean() < 1e-2, (step, k, got, v)
    cm_ctx.check_families(kind)
    # parameters after two AdamW steps
    for n, p in pm.state_dict().items():
        a, b = p.detach().cpu().double(), osd[n].double()
        if n.endswith("num_batches_tracked"):
            assert int(a) == int(b) == 2 * (2 if hier else 1)
            continue
        # two steps at lr=1e-4: an element whose gradient is pure rounding noise may move by
        # +-lr per step in either evaluation, i.e. differ by up to 4e-4
        # BN running statistics of the second step are taken on activations of the perturbed weights
        if "running_" in n:
            # Statistics of the second step are taken on activations of weights that AdamW's first update moved by +-lr per
            # element -- the sign of an element whose gradient is rounding noise differs between any two evaluations -- and the
            # lowest-resolution branch normalises over 8 samples here.  The yardstick is therefore the oracle's own two steps in
            # fp64: the product must be as close to them (L2) as the fp32 oracle is (x3), or within 1e-2 outright.  With EVERY
            # contraction of this 64-pixel net forced onto the 22-bit fp16x2 kernels ("fp16x2", "auto_ws") the factor is 6:
            # tests/test_grad_noise_gpu.py measures those modes 2-3.5x further from fp64 than the fp32 reference on this net
            # (observed here: 3.1x on stage4.0.branches.3.1.bn2.running_mean, fp16x2); the default routing keeps such small
            # layers on the exact-fp32 kernels.
            ref = osd64[n]
            e_prod = float((a - ref).norm()) / (float(ref.norm()) + 1e-4)
            e_orc = float((b - ref).norm()) / (float(ref.norm()) + 1e-4)
            assert e_prod < max(1e-2, (6 if mode in ("fp16x2", "auto_ws") else 3) * e_orc), (n, e_prod, e_orc)
            continue
        assert float((a - b).abs().max()) < 4.5e-4 + 1e-3 * float(b.abs().max()), n


def test_train_steps_track_the_oracle_at_256():
    """the same two steps on HRNet at 256x256, where the lowest-resolution branch still has 8 x 8 pixels (BN over 128
    samples): no chaos to excuse, so loss, per-class metrics, BN running statistics and the weights after two AdamW
    steps are held tightly -- in the deterministic mode, so that the comparison has no run-to-run noise of its own"""
    from oracle import models as OM
    from oracle import train_step as OT
    from hrseg_amd import _lib
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    from hrseg_amd.utils import synth
    kind, hier, tree_file, size, batch = "hrnet", True, "class_tree_tl.json", 256, 2
    tree = load_tree(tree_file)
    num_classes = [4, 4]
    weights = level_weights_for(tree_file, hier)
    args = _args(kind, hier, num_classes, weights, batch)
    xn, tn = synth.synthetic_batch(tree, batch, size, seed=23, hierarchical=True, blob=16)
    x, target = torch.from_numpy(xn), torch.from_numpy(tn)
    om = build_model(OM, kind, hier, tree, size)
    oopt = torch.optim.AdamW(om.parameters(), lr=1e-4)
    pm = build_model(PM, kind, hier, tree, size).cuda()
    popt = PT.FusedAdamW(pm, lr=[1e-4])
    loss_fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in num_classes]
    pm.train()
    _lib.set_deterministic(True)
    try:
        for step in range(2):
            ref = OT.train_step(om, oopt, x, target, num_classes, weights, hierarchical=hier, is_unet=False)
            loss, cms = PT.train_step(pm, popt, x.cuda(), target.cuda(), loss_fns, args, tree, [])
            assert abs(float(loss) - ref["loss"].item()) < TOL * abs(ref["loss"].item()), f"step {step}"
            vec = PT._metric_vectors(cms)
            for k, v in ref["metrics"].items():
                got = vec[k].cpu().numpy()
                # step 0: the same weights on both sides.  Step 1: Adam's first update moves EVERY element by +-lr
                # whatever its gradient's magnitude, so elements whose gradient is rounding noise (sign differs
                # between the two evaluations) end up 2 lr apart; on a random-init net that flips ~0.3 % of the
                # boundary pixels (observed: up to 3.9e-3 on one class, 1e-3 typical)
                tol = 2e-3 if step == 0 else 7e-3
                assert np.allclose(got, v, atol=tol) and np.abs(got - v).mean() < tol / 3, (step, k, got, v)
    finally:
        _lib.set_deterministic(False)
    osd = om.state_dict()
    for n, p in pm.state_dict().items():
        a, b = p.detach().cpu().double(), osd[n].double()
        if n.endswith("num_batches_tracked"):
            assert int(a) == int(b) == 4
            continue
        if "running_" in n:
            # (statistics of the second step are taken on activations of weights that differ by up to 2 lr: 1e-3 typical,
            # 3.4e-3 observed on the 8 x 8 branch; the first step's statistics are pinned at 1e-3 by the golden tests)
            assert float((a - b).norm()) < 1e-2 * float(b.norm()) + 1e-5, n
            continue
        # (an element whose gradient is rounding noise may move by +-lr per step in either evaluation)
        assert float((a - b).abs().max()) < 4.5e-4 + 1e-3 * float(b.abs().max()), n


def test_missing_library_is_loud(tmp_path, monkeypatch):
    """the product path has no CPU fallback: CPU tensors are rejected"""
    from hrseg_amd.Models import models as PM
    tree = load_tree("class_tree_tl.json")
    m = PM.UNet(size=32, n_channels=3, hierarchy=tree, model_type=1)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32), type=1)


@pytest.mark.parametrize("name", ["unet_hier_tl_62", "hrnet_hier_tl_64"])
def test_graphed_train_step_tracks_eager(name):
    """the hipGraph-replayed step (train.GraphedTrainStep) gives the eager step's losses (UNet: sequential level
    passes, HRNet: batched level passes)"""
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    num_classes = [int(v) for v in g["num_classes"]]
    weights = level_weights_for(tree_file, hier)
    args = _args(kind, hier, num_classes, weights, batch)
    x, target = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()

    def make():
        m = build_model(PM, kind, hier, tree, size).cuda()
        m.train()
        return m, PT.FusedAdamW(m, lr=[1e-4]), [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in num_classes]

    m, opt, fns = make()
    eager = [float(PT.train_step(m, opt, x, target, fns, args, tree, [])[0]) for _ in range(4)]
    m, opt, fns = make()
    graphed = PT.GraphedTrainStep(m, opt, fns, args, tree, x, target, warmup=1)   # runs eager step 0
    got = [float(graphed(x, target)[0]) for _ in range(3)]
    # UNet: the trajectories stay within 2e-3.  HRNet at 64x64 (2x2-pixel lowest branch, BN over 8 samples) is
    # chaotic: three EAGER runs from the same weights differ by 3e-4 / 2.5e-3 / 3e-3 at steps 2 / 3 / 4 (atomics
    # reorder sums), so later replays are only required to track loosely; the first replay must match.
    # (first replay: 6.7e-4 was observed inside a full-suite run, 3e-4 is typical; the deterministic variant below
    # is the exact statement)
    tols = [2e-3, 2e-3, 2e-3] if kind == "unet" else [1.5e-3, 1e-2, 3e-2]
    for a, b, tol in zip(got, eager[1:], tols):
        assert abs(a - b) < tol * abs(b), (got, eager)
    sd = opt.state_dict()                       # torch.optim.AdamW's layout; replays advance the device step counter
    assert all(float(st["step"]) == 4.0 for st in sd["state"].values()) and len(sd["state"]) == len(list(m.parameters()))


def test_graphed_train_step_matches_eager_in_deterministic_mode():
    """with the single-adder reductions (hrseg_tune deterministic) the chaotic 64x64 HRNet case has no noise to hide
    behind: three replays of the captured step must give the eager step's losses"""
    from hrseg_amd import _lib
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    name = "hrnet_hier_tl_64"
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    num_classes = [int(v) for v in g["num_classes"]]
    args = _args(kind, hier, num_classes, level_weights_for(tree_file, hier), batch)
    x, target = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()

    def make():
        m = build_model(PM, kind, hier, tree, size).cuda()
        m.train()
        return m, PT.FusedAdamW(m, lr=[1e-4]), [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in num_classes]

    _lib.set_deterministic(True)
    try:
        m, opt, fns = make()
        eager = [float(PT.train_step(m, opt, x, target, fns, args, tree, [])[0]) for _ in range(4)]
        m, opt, fns = make()
        graphed = PT.GraphedTrainStep(m, opt, fns, args, tree, x, target, warmup=1)
        got = [float(graphed(x, target)[0]) for _ in range(3)]
    finally:
        _lib.set_deterministic(False)
    for a, b in zip(got, eager[1:]):
        assert abs(a - b) <= 1e-6 * abs(b), (got, eager)


def test_consistency_on_model_probabilities_is_differentiable():
    """gradient through probs_per_level (composition + FiLM chain) against the oracle's autograd"""
    from oracle import models as OM
    from oracle import losses as OL
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    kind, hier, tree_file, size, batch = CASES["unet_hier_ext_32"]
    g = load_golden("unet_hier_ext_32")
    tree = load_tree(tree_file)
    x = torch.from_numpy(g["x"])
    om = build_model(OM, kind, hier, tree, size)
    om.train()
    probs, _ = om(x, type=1)
    lo = OL.hierarchical_consistency_loss(probs, om.levels, om.parent_of) + sum((p * p).mean() for p in probs)
    lo.backward()
    pm = build_model(PM, kind, hier, tree, size).cuda()
    pm.train()
    pprobs, _ = pm(x.cuda(), type=1)
    lp = PL.hierarchical_consistency_loss(pprobs, pm.levels, pm.parent_of) + sum((p * p).mean() for p in pprobs)
    assert abs(float(lp) - float(lo)) < 1e-4 * max(1.0, abs(float(lo)))
    lp.backward()
    og = dict(om.named_parameters())
    checked = 0
    for n, p in pm.named_parameters():
        if n.split(".")[0] in ("heads", "films"):
            ref = og[n].grad
            assert float((p.grad.cpu() - ref).abs().max()) < 5e-3 * float(ref.abs().max()) + 1e-7, n
            checked += 1
    assert checked >= 8


def test_train_epoch_and_test_loops_run():
    """the reference-shaped loops (train.py:161-393) on a synthetic loader"""
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd.Metrics import performance_metrics as PP
    from hrseg_amd import train as PT
    from hrseg_amd.utils.hierarchy import get_classes
    tree = load_tree("class_tree_tl.json")
    nc = get_classes(tree, full=True)
    weights = level_weights_for("class_tree_tl.json", True)
    args = _args("unet", True, nc, weights, 2)
    model = build_model(PM, "unet", True, tree, 32).cuda()
    opt = PT.FusedAdamW(model, lr=[1e-4])
    fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in nc]
    loader = PT.synthetic_loader(tree, 6, 32, 2, hierarchical=True, seed=1)
    mets = [PP.Accuracy(), PP.Jaccardindex(), PP.DiceScore(), PP.Precision(), PP.Recall()]
    out = PT.train_epoch(model, torch.device("cuda"), loader, opt, 1, fns, args, tree, None, *mets, epoch_num=1)
    loss, cls, acc, iou, dice, prec, rec, lvl = out
    assert np.isfinite(loss) and len(cls) == sum(nc) and 0 <= iou <= 1 and len(lvl) == 2
    res = PT.test(model, torch.device("cuda"), loader, 1, *mets, args, None, fns, tree, None)
    assert np.isfinite(res[0]) and len(res[2]) == sum(nc) and np.isfinite(res[-1])
    # the API-compatible get_metrics with the five metric objects agrees with the fused counts
    x, t = next(iter(loader))
    model.eval()
    with torch.no_grad():
        probs, logits = model(x.cuda(), type=1)
    targets = PT.split_targets(t.cuda(), args)
    acc_l, iou_l, dice_l, prec_l, rec_l = [], [], [], [], []
    cm = PT._new_class_metrics(sum(nc))
    PT.get_metrics(probs, targets, acc_l, iou_l, dice_l, prec_l, rec_l, mets[0], mets[1], mets[2], mets[3], mets[4],
                   torch.device("cuda"), cm, args)
    from hrseg_amd import ops
    cms = [ops.predict_metrics(p, tt, child=(L > 0), mask_pred=False, want_onehot=False)[1]
           for L, (p, tt) in enumerate(zip(probs, targets))]
    vec = PT._metric_vectors(cms)
    assert abs(iou_l[0] - float(vec["iou"].mean())) < 1e-6


@pytest.mark.parametrize("name", ["unet_hier_ext_32", "hrnet_hier_tl_64"])
def test_batched_and_dedup_passes_equal_sequential_passes(name):
    """opt-in `dedup_passes`: one backbone pass + L running-stat updates + summed head gradients must give
    what the L faithfully re-executed passes give: logits, loss, BN buffers (incl. num_batches_tracked = L)
    and every parameter gradient (fp32 summation order is the only difference)."""
    from hrseg_amd.Models import models as PM
    from hrseg_amd import train as PT
    from hrseg_amd.Metrics import losses as PL
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    x, target = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()
    num_classes = [int(v) for v in g["num_classes"]]
    weights = level_weights_for(tree_file, hier)
    args = _args(kind, hier, num_classes, weights, batch)
    results = []
    for mode in ("sequential", "dedup", "sequential", "batched"):
        model = build_model(PM, kind, hier, tree, size).cuda()
        model.dedup_passes = mode == "dedup"
        model.sequential_passes = mode == "sequential"
        model.train()
        probs, logits = PT._model_call(model, x, args, tree)
        targets = PT.split_targets(target, args)
        loss = 0.0
        for L, (z, t) in enumerate(zip(logits, targets)):
            ce, dice = PL.fused_ce_dice(z, t, weights[L])[:2]
            loss = loss + ce + dice
        loss = loss + PL.hierarchical_consistency_loss(probs, model.levels, model.parent_of)   # gradient into probs too
        loss.backward()
        results.append(dict(logits=[z.detach().cpu().numpy() for z in logits], loss=float(loss),
                            grads={n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()},
                            bufs={n: b.detach().cpu().numpy() for n, b in model.named_buffers()}))
    ref, ded, ref2, bat = results     # ref2: a second sequential run = the run-to-run noise of the atomics
    assert abs(ref["loss"] - ded["loss"]) < 1e-5 * max(1.0, abs(ref["loss"]))
    for a, b in zip(ref["logits"], ded["logits"]):
        assert rel_err(b, a) < 1e-5
    n_levels = len(ref["logits"])
    for n, b in ref["bufs"].items():
        if n.endswith("num_batches_tracked"):
            assert int(b) == n_levels and int(ded["bufs"][n]) == n_levels, n
        else:
            # sequential updates with the same statistics; split-K atomics make conv outputs (hence the
            # statistics) differ in the last bits from pass to pass
            assert rel_err(ded["bufs"][n], b) < 1e-5, n
    def worst_of(other):
        w, wn = 0.0, None
        for n, ga in ref["grads"].items():
            e = float(np.abs(ga - other["grads"][n]).max()) / max(float(np.abs(ga).max()), 1e-6)
            if e > w:
                w, wn = e, n
        return w, wn
    worst, worst_name = worst_of(ded)
    worst_noise, noise_name = worst_of(ref2)
    print(f"dedup vs faithful: worst {worst:.3e} ({worst_name}); faithful vs faithful: {worst_noise:.3e} ({noise_name})")
    # both are fp32 evaluations of the same sums in different order; early layers sit on the ~1e-2 fp32
    # noise floor of this net (tests/diagnostics/grad_noise.py), so the bound is that floor, not rounding
    assert worst < max(1e-1, 4 * worst_noise), (worst, worst_name, worst_noise)
    med = np.median([rel_err(ded["grads"][n], ga) for n, ga in ref["grads"].items() if np.abs(ga).max() > 0])
    noise = np.median([rel_err(ref2["grads"][n], ga) for n, ga in ref["grads"].items() if np.abs(ga).max() > 0])
    assert med < max(2e-2, 4 * noise), (med, noise)

    # the default execution: the L passes batched into one launch per layer -- same checks
    assert abs(ref["loss"] - bat["loss"]) < 1e-4 * max(1.0, abs(ref["loss"]))
    # (other tile plans / split-K choices at L*B images and other statistics chunking reorder fp32 sums)
    for a, b in zip(ref["logits"], bat["logits"]):
        assert rel_err(b, a) < 1e-4
    for n, b in ref["bufs"].items():
        if n.endswith("num_batches_tracked"):
            assert int(bat["bufs"][n]) == n_levels, n
        else:
            assert rel_err(bat["bufs"][n], b) < 1e-4, n
    worst_b, worst_b_name = worst_of(bat)
    med_b = np.median([rel_err(bat["grads"][n], ga) for n, ga in ref["grads"].items() if np.abs(ga).max() > 0])
    print(f"batched vs sequential: worst {worst_b:.3e} ({worst_b_name}), median {med_b:.3e}")
    assert worst_b < max(1e-1, 4 * worst_noise), (worst_b, worst_b_name, worst_noise)
    assert med_b < max(2e-2, 4 * noise), (med_b, noise)


@pytest.mark.parametrize("kind", ["hrnet", "unet"])
def test_full_size_step_properties(kind):
    """BASELINE.json's headline configuration (hier HRNet-W48, 620x620, batch 4: configs[2]) and configs[1] (hier UNet,
    620x620, batch 4) are too large for the CPU oracle inside a test; the step is checked through size-independent
    properties of the path: composition (children of a group sum to the parent's probability, level 0 is a sigmoid),
    masked predictions, confusion-matrix bookkeeping, finite loss/gradients, BN bookkeeping of the L passes."""
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT, ops
    from hrseg_amd.utils import synth
    tree = load_tree("class_tree_tl.json")
    model = build_model(PM, kind, True, tree, 620).cuda()
    x_np, t_np = synth.synthetic_batch(tree, 4, 620, seed=9, hierarchical=True)
    x, target = torch.from_numpy(x_np).cuda(), torch.from_numpy(t_np).cuda()
    weights = level_weights_for("class_tree_tl.json", True)
    args = _args(kind, True, [4, 4], weights, 4)
    model.train()
    probs, logits = PT._model_call(model, x, args, tree)
    assert [tuple(p.shape) for p in probs] == [(4, 4, 620, 620)] * 2
    p0, p1 = probs[0].detach(), probs[1].detach()
    assert float((p0 - torch.sigmoid(logits[0].detach())).abs().max()) < 1e-6
    # tl tree: the four level-1 classes are the children of 'tooth' (level-0 channel 3)
    assert float((p1.sum(1) - p0[:, 3]).abs().max()) < 1e-5
    assert float(p1.min()) >= 0.0 and float(p0.max()) <= 1.0
    targets = PT.split_targets(target, args)
    loss = 0.0
    for L, (z, t) in enumerate(zip(logits, targets)):
        onehot, cm = ops.predict_metrics(z.detach(), t, child=(L > 0), mask_pred=True)
        valid = (t != -1).all(1)
        assert float(onehot.sum(1)[valid].min()) == 1.0 and float(onehot.sum(1)[~valid].max() if (~valid).any() else 0) == 0.0
        # every valid pixel is counted once (child levels: pixels outside the parent carry the synthetic class 0)
        assert int(cm.sum()) == 4 * 620 * 620
        ce, dice = PL.fused_ce_dice(z, t, weights[L])[:2]
        assert torch.isfinite(ce) and torch.isfinite(dice)
        loss = loss + ce + dice
    loss.backward()
    total = 0.0
    for n, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        total += float(p.grad.double().pow(2).sum())
    assert total > 0.0
    for n, b in model.named_buffers():
        if n.endswith("num_batches_tracked"):
            assert int(b) == 2, n                  # two level passes -> two running-stat updates (D1)


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """save_checkpoint writes the reference's checkpoint dict (model_state_dict + torch.optim.AdamW-format
    optimizer_state_dict, train.py:668-703): a stock torch AdamW on the CPU oracle model loads it, and a
    fresh product model/optimizer resumed from the file continues exactly like the one that never stopped."""
    from oracle import models as OM
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    name = "unet_hier_tl_62"
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    num_classes = [int(v) for v in g["num_classes"]]
    weights = level_weights_for(tree_file, hier)
    args = _args(kind, hier, num_classes, weights, batch)
    x, target = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()
    loss_fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in num_classes]

    pm = build_model(PM, kind, hier, tree, size).cuda()
    popt = PT.FusedAdamW(pm, lr=[1e-3])
    pm.train()
    ll = []
    for _ in range(2):
        loss, _ = PT.train_step(pm, popt, x, target, loss_fns, args, tree, ll)
    path = str(tmp_path / "last.pt")
    PT.save_checkpoint(path, pm, popt, epoch=3, loss=float(loss), test_measure_mean=0.5, test_measure_std=0.1)

    # (1) the file is what the reference's loader expects
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss", "test_measure_mean", "test_measure_std"}
    om = build_model(OM, kind, hier, tree, size)
    om.load_state_dict(ck["model_state_dict"])
    oopt = torch.optim.AdamW(om.parameters(), lr=1e-3)
    oopt.load_state_dict(ck["optimizer_state_dict"])
    st = oopt.state_dict()["state"]
    params = list(om.parameters())
    assert len(st) == len(params) and all(float(st[i]["step"]) == 2.0 for i in st)
    for i, p in enumerate(params):
        assert st[i]["exp_avg"].shape == p.shape and st[i]["exp_avg_sq"].shape == p.shape
    flat = pm.flatten_parameters()
    some = [i for i, p in enumerate(pm.parameters()) if p.dim() == 4][:3]
    for i in some:
        p = list(pm.parameters())[i]
        assert torch.equal(st[i]["exp_avg"], flat.view_of(popt._m, p).cpu())

    # (2) resume: a fresh model + optimizer from the file takes the same third step
    pm2 = build_model(PM, kind, hier, tree, size).cuda()
    popt2 = PT.FusedAdamW(pm2, lr=[5e-2])                  # overwritten by the checkpoint's param_groups
    ck2 = PT.load_checkpoint(path, pm2, popt2)
    assert ck2["epoch"] == 3 and popt2.param_groups[0]["lr"] == 1e-3
    pm2.train()
    l1, _ = PT.train_step(pm, popt, x, target, loss_fns, args, tree, ll)
    l2, _ = PT.train_step(pm2, popt2, x, target, loss_fns, args, tree, [])
    assert abs(float(l1) - float(l2)) < 1e-5 * abs(float(l1))
    sd1, sd2 = pm.state_dict(), pm2.state_dict()
    for n in sd1:
        a, b = sd1[n].double(), sd2[n].double()
        # identical inputs and state; split-K / weight-gradient atomics reorder sums run to run, and Adam
        # turns a sign flip of a noise-level gradient into +-lr
        assert float((a - b).abs().max()) <= 2.1e-3 + 1e-4 * float(b.abs().max()), n

    # (3) a state_dict written by stock torch AdamW (= a checkpoint of the reference) loads as well
    popt3 = PT.FusedAdamW(pm2, lr=[1e-3])
    popt3.load_state_dict(oopt.state_dict())
    assert popt3._step == 2
    for i in some:
        p = list(pm2.parameters())[i]
        assert torch.equal(pm2.flatten_parameters().view_of(popt3._m, p).cpu(), st[i]["exp_avg"])


def test_train_steps_do_not_retain_memory():
    """a step must not keep its logits/probabilities alive (reference cycle through the autograd node)"""
    import gc
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    kind, hier, tree_file, size, batch = CASES["unet_hier_tl_62"]
    g = load_golden("unet_hier_tl_62")
    tree = load_tree(tree_file)
    num_classes = [int(v) for v in g["num_classes"]]
    args = _args(kind, hier, num_classes, level_weights_for(tree_file, hier), batch)
    x, target = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()
    model = build_model(PM, kind, hier, tree, size).cuda()
    opt = PT.FusedAdamW(model, lr=[1e-4])
    fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in num_classes]
    model.train()
    ll, seen = [], []
    for step in range(6):
        PT.train_step(model, opt, x, target, fns, args, tree, ll)
        torch.cuda.synchronize()
        gc.collect()
        seen.append(torch.cuda.memory_allocated())
    assert seen[2] == seen[3] == seen[4] == seen[5], seen
    # forward without a backward (e.g. a validation loss under grad mode) is released as well
    for _ in range(3):
        PT._model_call(model, x, args, tree)
        gc.collect()
        seen.append(torch.cuda.memory_allocated())
    assert seen[-1] == seen[-2] == seen[5], seen


@pytest.mark.parametrize("mode", CONV_MODES)
@pytest.mark.parametrize("kind,H,W,B", [("unet", 50, 66, 3), ("hrnet", 70, 44, 1), ("hrnet", 36, 100, 3)])
def test_ragged_shapes_against_the_oracle(kind, H, W, B, mode):
    """non-square, odd-quarter sizes and batch 1/3 (floor/pad paths of UNet, odd HRNet branch sizes, the
    batched level passes at an odd batch): train-mode logits, loss and eval-mode logits against the oracle"""
    from oracle import models as OM
    from oracle import losses as OL
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd.utils import synth
    tree = load_tree("class_tree_tl.json")
    weights = level_weights_for("class_tree_tl.json", True)
    g = np.random.Generator(np.random.PCG64(H * W + B))
    x = torch.from_numpy(g.standard_normal((B, 3, H, W)).astype(np.float32))
    lab = g.integers(0, 7, size=(B, H, W))
    target = torch.from_numpy(synth.encode_targets(lab, tree, True))
    key = (kind, H, W, B)
    if key not in _ORACLE_RAGGED:          # the oracle's passes do not depend on the product's convolution mode: run once
        om = build_model(OM, kind, True, tree, max(H, W))
        om.train()
        with torch.no_grad():
            _, zo = om(x, type=1) if kind == "unet" else om(x)
        om.eval()
        with torch.no_grad():
            _, ze = om(x, type=1) if kind == "unet" else om(x)
        _ORACLE_RAGGED[key] = ([z.clone() for z in zo], [z.clone() for z in ze])
    zo, zo_eval = _ORACLE_RAGGED[key]
    pm = build_model(PM, kind, True, tree, max(H, W)).cuda()
    pm.train()
    with conv_mode(pm, mode) as cm_ctx:
        _, zp = pm(x.cuda(), type=1) if kind == "unet" else pm(x.cuda())
    if mode in ("fp16x2", "auto_ws"):
        c = cm_ctx.counts          # (36 x 100: the 9 x 25 branch image is mostly tile padding for the wave-specialised body)
        assert c["ws"] + c["ws_group"] + c["patch_sp"] + c["sp_pgroup"] + c["sp_group"] + c["sp_im2col"] > 0, c
        assert (c["ws"] + c["ws_group"] > 0) or (H, W) == (36, 100), c
    lo = lp = 0.0
    for L, (a, b) in enumerate(zip(zo, zp)):
        assert rel_err(b.detach().cpu().numpy(), a.detach().numpy()) < TOL, f"train logits {L}"
        t = target[:, 4 * L:4 * L + 4]
        lo = lo + OL.cross_entropy_loss(a, t, logits_input=True, class_weight=weights[L])
        ce, dice = PL.fused_ce_dice(b, t.cuda(), weights[L])[:2]
        lp = lp + ce
    assert abs(float(lp) - float(lo)) < TOL * max(1.0, abs(float(lo)))
    pm.eval()
    with torch.no_grad():
        with conv_mode(pm, mode):
            _, zp = pm(x.cuda(), type=1) if kind == "unet" else pm(x.cuda())
    for L, (a, b) in enumerate(zip(zo_eval, zp)):
        assert rel_err(b.cpu().numpy(), a.numpy()) < TOL, f"eval logits {L}"


# ---------------------------------------------------------------------------------------------------------------
# SURVEY 8(f4) opt-in extensions (default off; the reference has none of them in running code)
# (the HRNet case's oracle side -- four CPU evaluations, one in fp64: 160 s -- is a committed fixture,
# tests/golden/concat_oracle_hrnet_64.npz, written by tests/golden/gen_f4_fixtures.py with tests/helpers.concat_oracle_results;
# the UNet case evaluates the oracle live)
@pytest.mark.parametrize("kind,size", [("unet", 64), ("hrnet", 64)])
def test_concat_prev_logits_against_the_oracle(kind, size):
    """logit-concatenated re-encoding (north_star wording; models.py:267,277 is where the reference re-runs on the image
    only): level L >= 1 encodes cat(image, logits_{L-1}) through its own first convolution.  Train-mode logits, loss,
    the gradients that only exist because of the concatenation (cond_stems, and the part of level 0's head gradient that
    flows back through level 1's input) and eval-mode logits against the oracle twin."""
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from tests.helpers import concat_inputs, concat_model, concat_oracle_results, load_concat_fixture
    tree = load_tree("class_tree_tl.json")
    weights = level_weights_for("class_tree_tl.json", True)
    o = load_concat_fixture(kind, size) or concat_oracle_results(kind, size)
    pm = concat_model(PM, kind, size, tree).cuda()
    pm.conv_dtype = "f32"
    assert list(o["param_names"]) == [n for n, _ in pm.named_parameters()]
    assert any(n.startswith("cond_stems.0.") for n, _ in pm.named_parameters())
    x, target = concat_inputs(tree, size)
    pm.train()
    _, zp = pm(x.cuda(), type=1) if kind == "unet" else pm(x.cuda())
    lp = 0.0
    for L, b in enumerate(zp):
        # level L >= 1 re-encodes level L-1's logits: their (in-tolerance) difference from the oracle's is an INPUT
        # perturbation of this pass on top of the pass's own rounding
        assert rel_err(b.detach().cpu().numpy(), o[f"train_logits{L}"]) < (TOL if L == 0 else 3 * TOL), f"train logits {L}"
        t = target[:, 4 * L:4 * L + 4]
        ce, dice = PL.fused_ce_dice(b, t.cuda(), weights[L])[:2]
        lp = lp + ce + dice
    lo = float(o["loss"])
    assert abs(float(lp) - lo) < TOL * max(1.0, abs(lo))
    lp.backward()
    head0 = "heads.0.conv.weight" if kind == "unet" else "classifiers.0.weight"
    # the first convolution sits behind every BN / ReLU of the net, where fp32 evaluations differ from each other at the
    # percent level (ReLU flips; tests/diagnostics/grad_noise.py).  The yardstick is therefore an fp64 evaluation of the
    # oracle: the product must be as close to it as the fp32 oracle is (x3), or within 5e-2 outright.
    for n, p in pm.named_parameters():
        if n.startswith("cond_stems."):
            ref = o["g64::" + n]
            scale = float(np.abs(ref).max())
            e_prod = float(np.abs(p.grad.cpu().double().numpy() - ref).max()) / scale
            e_orc = float(np.abs(o["g32::" + n].astype(np.float64) - ref).max()) / scale
            print(f"{n}: product vs fp64 {e_prod:.3e}, fp32 oracle vs fp64 {e_orc:.3e}")
            assert scale > 0 and e_prod < max(5e-2, 3 * e_orc), (n, e_prod, e_orc)
    # the gradient that exists only because of the concatenation: the oracle with level 1's input DETACHED from the logits
    # of level 0 gives a different gradient for level 0's head; the product must sit with the full one
    g_full, g_det = o["g64::" + head0].astype(np.float32), o["gdet::" + head0]
    g_prod = dict(pm.named_parameters())[head0].grad.cpu().numpy()
    gap = float(np.abs(g_full - g_det).max())
    err = float(np.abs(g_prod - g_full).max())
    print(f"head-0 gradient: |full - detached| = {gap:.3e}, |product - full| = {err:.3e}, max |full| = {float(np.abs(g_full).max()):.3e}")
    assert gap > 0 and err < 0.3 * gap, (gap, err)
    pm.eval()
    with torch.no_grad():
        _, zp = pm(x.cuda(), type=1) if kind == "unet" else pm(x.cuda())
    for L, b in enumerate(zp):
        assert rel_err(b.cpu().numpy(), o[f"eval_logits{L}"]) < (TOL if L == 0 else 3 * TOL), f"eval logits {L}"
    pm.train()
    pm.dedup_passes = True
    with pytest.raises(RuntimeError):
        pm(x.cuda(), type=1) if kind == "unet" else pm(x.cuda())
