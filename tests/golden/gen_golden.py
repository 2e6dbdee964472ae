"""Generate golden vectors from the REFERENCE itself (run in the build container only).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

Imports /root/reference/Models/models.py and Metrics/losses.py (never copied
into this repo).  Their module headers import packages that are not installed
offline and are never used on this path (timm `_cfg`, segmentation_models_pytorch,
torchvision, torchmetrics -- SURVEY.md section 8c); empty in-memory stand-ins
satisfy those import statements.  Weights come from the name-keyed recipe in
utils/synth.py, inputs from its seeded generators, so the fixtures hold only
inputs and expected outputs.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from hrseg_amd.utils import synth  # noqa: E402
from hrseg_amd.utils.config import hrnet_w48_config  # noqa: E402
from hrseg_amd.utils.hierarchy import get_classes  # noqa: E402

CASES = [
    # name, model, hierarchical, tree file, size, batch
    ("unet_flat_tl_32", "unet", False, "class_tree_tl.json", 32, 2),
    ("unet_hier_tl_62", "unet", True, "class_tree_tl.json", 62, 2),
    ("unet_hier_ext_32", "unet", True, "class_tree_tl_extended.json", 32, 2),
    ("hrnet_flat_tl_64", "hrnet", False, "class_tree_tl.json", 64, 2),
    ("hrnet_hier_tl_64", "hrnet", True, "class_tree_tl.json", 64, 2),
    ("hrnet_hier_ext_62", "hrnet", True, "class_tree_tl_extended.json", 62, 2),
]

EXT_WEIGHTS = [[0.3, 1.2], [0.8, 1.1], [1.5, 1.4, 2.0, 0.6], [1.6, 0.4, 1.0]]


def level_weights_for(tree_file, hierarchical):
    if not hierarchical:
        return synth.README_LEVEL_WEIGHTS_FLAT
    return synth.README_LEVEL_WEIGHTS_TL if tree_file == "class_tree_tl.json" else EXT_WEIGHTS


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub("timm")
    stub("timm.models")
    stub("timm.models.vision_transformer", _cfg=lambda **kw: {})
    stub("segmentation_models_pytorch")
    stub("torchvision")
    stub("torchmetrics")
    sys.path.insert(0, REF)
    from Models import models as ref_models
    from Metrics import losses as ref_losses
    return ref_models, ref_losses


def run_case(ref_models, ref_losses, name, kind, hier, tree_file, size, batch):
    tree = json.load(open(os.path.join(REF, tree_file)))
    torch.manual_seed(0)
    if kind == "unet":
        model = ref_models.UNet(size=size, n_channels=3, hierarchy=tree, model_type=1 if hier else 0)
    else:
        model = ref_models.HighResolutionNet(hrnet_w48_config(), hierarchy=tree, model_type=1 if hier else 0)
    synth.fill_state_dict(model)
    x_np, t_np = synth.synthetic_batch(tree, batch, size, seed=len(name), hierarchical=hier, blob=4)
    x, target = torch.from_numpy(x_np), torch.from_numpy(t_np)
    num_classes = get_classes(tree, full=hier)
    weights = level_weights_for(tree_file, hier)
    out = {"x": x_np, "target": t_np, "num_classes": np.array(num_classes if hier else [sum(num_classes)])}

    # eval-mode forward (running-stat BN)
    model.eval()
    with torch.no_grad():
        probs, logits = model(x, type=1 if hier else 0) if kind == "unet" else model(x)
    if not hier:
        logits = [logits]
    for L, z in enumerate(logits):
        out[f"eval_logits{L}"] = z.numpy()

    # train-mode forward + loss + backward, as train.py:198-241 (kwargs bug D3 dropped)
    model.train()
    probs, logits = model(x, type=1 if hier else 0) if kind == "unet" else model(x)
    if not hier:
        logits, targets = [logits], [target]
    else:
        targets, s = [], 0
        for n in num_classes:
            targets.append(target[:, s:s + n])
            s += n
    onehots = []
    for z, t in zip(logits, targets):
        oh = torch.nn.functional.one_hot(torch.argmax(torch.softmax(z, 1), 1), z.shape[1]).permute(0, 3, 1, 2).float()
        onehots.append(torch.where(t == -1, 0, oh))
    loss = 0.0
    for L, (z, t) in enumerate(zip(logits, targets)):
        ce = ref_losses.CrossEntropyLoss()(z, t, class_weight=weights[L], logits_input=True)
        dice = ref_losses.SoftDiceLoss()(z, t, class_weight=weights[L], logits_input=True)
        out[f"ce{L}"] = np.float32(ce.item())
        out[f"dice{L}"] = np.float32(dice.item()) if dice is not None else np.float32(np.nan)
        loss = loss + ce + (dice if dice is not None else 0.0)
        out[f"logits{L}"] = z.detach().numpy()
        out[f"onehot{L}"] = onehots[L].numpy()
    if hier:
        for L, p in enumerate(probs):
            out[f"probs{L}"] = p.detach().numpy()
        cons = ref_losses.hierarchical_consistency_loss(onehots, model.levels, model.parent_of)
        out["cons_onehot"] = np.float32(float(cons))
        out["cons_probs"] = np.float32(float(ref_losses.hierarchical_consistency_loss(
            [p.detach() for p in probs], model.levels, model.parent_of)))
        loss = loss + cons
    out["loss"] = np.float32(loss.item())
    loss.backward()

    names, norms = [], []
    for n, p in model.named_parameters():
        names.append(n)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    for n, p in model.named_parameters():
        head = n.split(".")[0]
        if head in ("heads", "films", "classifier", "classifiers", "out_flat") or n.startswith("stem.0") \
                or n.startswith("inc0.conv.conv.0"):
            out["grad::" + n] = p.grad.numpy()
    bn_names, bn_vals = [], []
    for n, b in model.named_buffers():
        bn_names.append(n)
        bn_vals.append(float(b.double().norm()))
    out["buf_names"] = np.array(bn_names)
    out["buf_norms"] = np.array(bn_vals, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={out['loss']:.6f} params={len(names)}")


def loss_only_cases(ref_losses):
    """Loss edge cases: empty masks (NaN rules), all-ignored items, random logits."""
    g = np.random.Generator(np.random.PCG64(7))
    out = {}
    B, C, S = 3, 4, 16
    z = g.standard_normal((B, C, S, S)).astype(np.float32) * 2
    t = g.integers(-1, 2, size=(B, C, S, S)).astype(np.float32)
    t[1] = -1.0                      # item 1 fully ignored -> CE item 1.0, Dice item dropped
    t[2, 3] = -1.0                   # one empty class mask -> CE item NaN -> 1.0
    w = [0.5, 1.5, 1.0, 2.0]
    zt = torch.from_numpy(z).requires_grad_(True)
    ce = ref_losses.CrossEntropyLoss()(zt, torch.from_numpy(t), class_weight=w, logits_input=True)
    dice = ref_losses.SoftDiceLoss()(zt, torch.from_numpy(t), class_weight=w, logits_input=True)
    (ce + dice).backward()
    out.update(z=z, t=t, w=np.array(w, np.float32), ce=np.float32(ce.item()), dice=np.float32(dice.item()),
               dz=zt.grad.numpy())
    t_all = np.full((2, C, 8, 8), -1.0, np.float32)
    z_all = g.standard_normal((2, C, 8, 8)).astype(np.float32)
    ce2 = ref_losses.CrossEntropyLoss()(torch.from_numpy(z_all), torch.from_numpy(t_all), class_weight=w, logits_input=True)
    d2 = ref_losses.SoftDiceLoss()(torch.from_numpy(z_all), torch.from_numpy(t_all), class_weight=w, logits_input=True)
    out.update(z_all=z_all, t_all=t_all, ce_all=np.float32(ce2.item()), dice_all_is_none=np.array(d2 is None))
    np.savez_compressed(os.path.join(HERE, "loss_cases.npz"), **out)
    print("loss_cases: ce=%.6f dice=%.6f ce_all=%.3f dice_all_none=%s" % (out["ce"], out["dice"], out["ce_all"], d2 is None))


def main():
    ref_models, ref_losses = import_reference()
    torch.set_num_threads(8)
    loss_only_cases(ref_losses)
    for c in CASES:
        run_case(ref_models, ref_losses, *c)


if __name__ == "__main__":
    main()
