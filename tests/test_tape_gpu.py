"""The recorded launch tape of the train step (train.TapedTrainStep, the default way train_epoch and bench.py issue a step)
against the eager step it was recorded from: same kernels, same arguments, same order -- in deterministic mode the same BITS
(losses, confusion counts, weights, BN buffers, optimizer moments) over several steps on changing batches; in the default mode
the same values to atomic-order noise.  Also: the recorded body contains no ATen launch (a launch the tape would silently
drop), folded inference weights follow a replay, and train_epoch gives the same epoch through either path."""
import argparse
import os

import numpy as np
import pytest
import torch

from tests.helpers import CASES, build_model, level_weights_for, load_golden, load_tree

pytestmark = pytest.mark.gpu


def _setup(name, lr=1e-3):
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    nc = [int(v) for v in g["num_classes"]]
    args = argparse.Namespace(model_type=1 if hier else 0, model_select=0 if kind == "unet" else 1, num_classes=nc,
                              level_weights=level_weights_for(tree_file, hier), level0_pretrain_epochs=None, batch_size=batch)
    model = build_model(PM, kind, hier, tree, size).cuda()
    model.train()
    opt = PT.FusedAdamW(model, lr=[lr])
    fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in nc]
    return model, opt, fns, args, tree, g


def _batches(g, n):
    """n different batches of the golden's shape (the golden batch, then seeded perturbations of it)"""
    x0, t0 = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()
    out = [(x0, t0)]
    gen = torch.Generator(device="cuda").manual_seed(5)
    for i in range(1, n):
        out.append((x0 + 0.1 * torch.randn(x0.shape, generator=gen, device="cuda"), t0.roll(i, dims=-1).contiguous()))
    return out


@pytest.mark.parametrize("name", ["hrnet_hier_tl_64", "unet_hier_tl_62", "hrnet_flat_tl_64"])
def test_tape_replay_is_bit_identical_to_the_eager_step_in_deterministic_mode(name):
    from hrseg_amd import _lib, train as PT
    _lib.set_deterministic(True)
    try:
        batches = None
        results = {}
        for mode in ("eager", "tape"):
            model, opt, fns, args, tree, g = _setup(name)
            batches = batches or _batches(g, 4)
            losses, cms_all, ll = [], [], []
            taped = None
            for i, (x, t) in enumerate(batches):
                if mode == "eager":
                    loss, cms = PT.train_step(model, opt, x, t, fns, args, tree, ll)
                    losses.append(float(loss))
                else:
                    if taped is None:
                        taped = PT.TapedTrainStep(model, opt, fns, args, tree, x, t)
                        packed, cms = taped.result()
                    else:
                        packed, cms = taped(x, t)
                    losses.append(taped.unpack(packed.tolist())[0])
                cms_all.append([c.cpu().clone() for c in cms])
            torch.cuda.synchronize()
            if mode == "tape":
                assert taped.replays == len(batches) - 1 and taped.tape.calls > 50
            results[mode] = (losses, cms_all, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                             opt._m.cpu().clone(), opt._v.cpu().clone(), int(opt._state[0].item()))
        (le, ce, se, me, ve, ne), (lt, ct, st, mt, vt, nt) = results["eager"], results["tape"]
        assert ne == nt == len(batches)
        assert le == lt, (le, lt)                       # the host-side arithmetic of unpack() is get_loss's, bit for bit
        for a, b in zip(ce, ct):
            for u, v in zip(a, b):
                assert torch.equal(u, v)
        for k in se:
            assert torch.equal(se[k], st[k]), k
        assert torch.equal(me, mt) and torch.equal(ve, vt)
    finally:
        _lib.set_deterministic(False)


def test_tape_replay_tracks_the_eager_step_in_the_default_mode():
    """default (atomics on): three steps through either path agree to the run-to-run noise of the eager step itself"""
    from hrseg_amd import train as PT
    res = {}
    for mode in ("eager", "eager2", "tape"):
        model, opt, fns, args, tree, g = _setup("hrnet_hier_tl_64", lr=1e-4)
        bs = _batches(g, 3)
        losses, taped, ll = [], None, []
        for x, t in bs:
            if mode != "tape":
                losses.append(float(PT.train_step(model, opt, x, t, fns, args, tree, ll)[0]))
            elif taped is None:
                taped = PT.TapedTrainStep(model, opt, fns, args, tree, x, t)
                losses.append(taped.unpack(taped.result()[0].tolist())[0])
            else:
                losses.append(taped.unpack(taped(x, t)[0].tolist())[0])
        res[mode] = losses
    noise = max(abs(a - b) for a, b in zip(res["eager"], res["eager2"]))
    for a, b in zip(res["eager"], res["tape"]):
        # (the 64-pixel case is chaotic from the second step on: two eager runs differ by up to ~1e-3 of the loss, and their
        # difference in one sample of three steps can come out several times smaller than that)
        assert abs(a - b) <= max(1e-3 * abs(a), 4 * noise), (res, noise)


def test_recorded_body_launches_no_aten_kernel():
    """a kernel torch launches inside the recorded body would be missing from every replay: the body must consist of
    C-ABI calls only.  One more pass of the body under the profiler: every device activity is one of the library's kernels."""
    from torch.profiler import ProfilerActivity, profile
    from hrseg_amd import train as PT
    model, opt, fns, args, tree, g = _setup("hrnet_hier_tl_64")
    x, t = _batches(g, 1)[0]
    taped = PT.TapedTrainStep(model, opt, fns, args, tree, x, t)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        with torch.no_grad():
            taped._body()
        torch.cuda.synchronize()
    names = [ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA]
    assert len(names) > 100
    foreign = [n for n in names if "at::" in n or "Memcpy" in n or "Memset" in n or "elementwise_kernel" in n]
    assert not foreign, sorted(set(foreign))


@pytest.mark.parametrize("kind", ["tape", "graph"])
def test_folded_inference_weights_follow_replayed_steps(kind):
    """eval, replay k steps (no host-side optimizer.step / model forward runs), eval -- the second eval must see
    the new weights and running statistics: compared with the unfolded inference path (fold_bn = False)"""
    from hrseg_amd import train as PT
    model, opt, fns, args, tree, g = _setup("unet_hier_tl_62", lr=1e-2)
    x, t = _batches(g, 1)[0]

    def eval_logits(fold):
        model.eval()
        model.fold_bn = fold
        with torch.no_grad():
            _, z = PT._model_call(model, x, args, tree)
        model.fold_bn = True
        model.train()
        return [v.clone() for v in z]

    before = eval_logits(True)
    step = PT.TapedTrainStep(model, opt, fns, args, tree, x, t) if kind == "tape" else \
        PT.GraphedTrainStep(model, opt, fns, args, tree, x, t, warmup=1)
    for _ in range(3):
        step(x, t)
    folded, plain = eval_logits(True), eval_logits(False)
    for a, b, c in zip(folded, plain, before):
        assert float((a - b).abs().max()) < 1e-3 * float(b.abs().max())
        assert float((a - c).abs().max()) > 1e-2 * float(c.abs().max())        # the steps did move the outputs


def test_train_epoch_is_the_same_epoch_with_and_without_the_tape(monkeypatch):
    from hrseg_amd import _lib, train as PT
    from hrseg_amd.Metrics import performance_metrics as PP
    _lib.set_deterministic(True)
    try:
        out = {}
        for tape in ("1", "0"):
            monkeypatch.setenv("HRSEG_TAPE", tape)
            model, opt, fns, args, tree, g = _setup("unet_hier_tl_62")
            loader = PT.synthetic_loader(tree, 6, 62, 2, hierarchical=True, seed=3)
            mets = [PP.Accuracy(), PP.Jaccardindex(), PP.DiceScore(), PP.Precision(), PP.Recall()]
            r = PT.train_epoch(model, torch.device("cuda"), loader, opt, 1, fns, args, tree, None, *mets, 1)
            out[tape] = (r, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                         len(getattr(model, "_hr_tapes", {})))
        (ra, sa, na), (rb, sb, nb) = out["1"], out["0"]
        assert na == 1 and nb == 0
        assert ra[0] == rb[0] and ra[2:7] == rb[2:7]
        assert np.allclose(ra[7], rb[7], rtol=1e-6, atol=0)
        for ma, mb in zip(ra[1], rb[1]):
            assert ma == mb
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    finally:
        _lib.set_deterministic(False)


@pytest.mark.parametrize("name", ["hrnet_hier_tl_64", "unet_hier_tl_62"])
def test_cached_weight_images_follow_the_weights(name, monkeypatch):
    """persistent weight images (hrseg_set_weight_image_arena: one refresh launch per model call instead of an image launch in
    front of every convolution) against the per-convolution images (HRSEG_WEIGHT_IMAGES=0): three steps at a LARGE learning
    rate, so that a stale image would show at once -- same bits in deterministic mode, eager and taped"""
    from hrseg_amd import _lib, train as PT
    from tests.helpers import conv_mode
    _lib.set_deterministic(True)
    try:
        res = {}
        for images, taped in (("1", False), ("0", False), ("1", True)):
            monkeypatch.setenv("HRSEG_WEIGHT_IMAGES", images)
            model, opt, fns, args, tree, g = _setup(name, lr=3e-2)
            bs = _batches(g, 3)
            losses, step = [], None
            with conv_mode(model, "auto_ws") as cm:          # (the wave-specialised kernels at golden sizes)
                for x, t in bs:
                    if not taped:
                        losses.append(float(PT.train_step(model, opt, x, t, fns, args, tree, [])[0]))
                    elif step is None:
                        step = PT.TapedTrainStep(model, opt, fns, args, tree, x, t)
                        losses.append(step.unpack(step.result()[0].tolist())[0])
                    else:
                        losses.append(step.unpack(step(x, t)[0].tolist())[0])
                torch.cuda.synchronize()
            assert cm.counts["ws"] + cm.counts["ws_group"] > 0
            res[(images, taped)] = (losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        base = res[("0", False)]
        for key in (("1", False), ("1", True)):
            assert res[key][0] == base[0], (key, res[key][0], base[0])
            for k, v in base[1].items():
                assert torch.equal(res[key][1][k], v), (key, k)
        assert abs(base[0][0] - base[0][2]) > 1e-3 * abs(base[0][0])       # the steps did move the loss
    finally:
        _lib.set_deterministic(False)


@pytest.mark.parametrize("name", ["hrnet_hier_tl_64", "unet_hier_tl_62"])
def test_presplit_activations(name, monkeypatch):
    """Pre-split activations (hrseg_bn_fwd_t.z_split / hrseg_conv_shape_t.x_split).  Level 1 -- conv1 -> conv2 of a block, a
    tensor with no other reader: BatchNorm writes the bytes conv2's staging would compute from the fp32 tensor, so three train
    steps give the SAME BITS as level 0 (deterministic mode).  Level 2 (default) -- also a block's output read by the next block
    of the branch only: that block adds it as the residual in its 22-bit form, a 2^-23 relative rounding per block: losses
    within 2e-6, weights within Adam's +-lr sign noise."""
    from hrseg_amd import _lib, engine, train as PT
    from tests.helpers import conv_mode
    _lib.set_deterministic(True)
    try:
        res = {}
        for level in (0, 1, 2):
            monkeypatch.setattr(engine, "_X_SPLIT", level)
            model, opt, fns, args, tree, g = _setup(name, lr=1e-3)
            nsplit, nres = [0], [0]
            orig = engine.ops.bn_fwd_group

            def spy(items, *a, **k):
                nsplit[0] += sum(bool(it.get("z_split")) for it in items)
                nres[0] += sum(bool(it.get("residual_split")) for it in items)
                return orig(items, *a, **k)
            monkeypatch.setattr(engine.ops, "bn_fwd_group", spy)
            losses = []
            with conv_mode(model, "auto_ws"):
                for x, t in _batches(g, 3):
                    losses.append(float(PT.train_step(model, opt, x, t, fns, args, tree, [])[0]))
                torch.cuda.synchronize()
            monkeypatch.setattr(engine.ops, "bn_fwd_group", orig)
            res[level] = (losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, nsplit[0], nres[0])
        assert res[0][2] == 0 and res[1][2] > 0 and res[1][3] == 0
        assert res[1][0] == res[0][0], (res[1][0], res[0][0])
        for k, v in res[0][1].items():
            assert torch.equal(res[1][1][k], v), k
        if name.startswith("hrnet"):
            assert res[2][2] > res[1][2] and res[2][3] > 0          # block outputs split, read back as residuals
            # first step: same weights, the rounding alone; later steps: AdamW turns a sign flip of a noise-level gradient
            # element into 2 lr (observed 2e-4 relative on the loss at lr = 1e-3)
            assert abs(res[2][0][0] - res[0][0][0]) < 2e-6 * abs(res[0][0][0]) + 1e-7, (res[2][0], res[0][0])
            for a, b in zip(res[2][0][1:], res[0][0][1:]):
                assert abs(a - b) < 1e-3 * abs(b), (res[2][0], res[0][0])
            for k, v in res[0][1].items():
                if not v.dtype.is_floating_point:
                    continue
                if "running_" in k:      # statistics of steps 2 and 3 are taken on activations of weights +-lr apart (64-pixel net:
                    d = res[2][1][k] - v       # the lowest branch normalises over 8 samples); compared in the L2 sense
                    assert float(d.norm()) < 1e-1 * float(v.norm()) + 1e-4, k          # (observed up to 3e-2 on a 4 x 4 fuse layer)
                else:
                    assert float((res[2][1][k] - v).abs().max()) < 6.5e-3 + 1e-3 * float(v.abs().max()), k     # 3 steps x +-lr x 2
    finally:
        _lib.set_deterministic(False)
