"""Headline-size fixtures from the REFERENCE itself (run in the build container only, about ten minutes of CPU).

    python tests/golden/gen_golden_620.py [name ...]      # writes tests/golden/<name>.npz

BASELINE.json's headline geometry (620x620, class_tree_tl.json, hierarchical) at batch 2, for HRNet-W48 (configs[2]) and
UNet (configs[1]): the tile plans that exist only at this size (canvas tiling of the 20 / 39-pixel branches, the wide im2col
and wide weight-gradient bodies, four-branch wave-specialised groups with the real block partition) meet the reference here.
Full tensors at this size would be hundreds of MB, so a fixture holds
  * nothing of the inputs but their seed and a checksum (both sides rebuild them with utils/synth.py),
  * per-level CE / Dice, consistency, total loss,
  * train- and eval-mode logits and train-mode probabilities on a stride-5 pixel lattice (124 x 124 points per plane),
  * the arg-max class histogram and the confusion counts of the one-hot prediction prep (train.py:206-231) per level,
  * per-parameter gradient L2 norms and one seeded +-1 projection per parameter (sign-sensitive), both from the fp32
    reference AND from the reference evaluated in fp64 -- the fp64 values are the yardstick (tests/test_grad_noise_gpu.py):
    the product must be as close to them as the fp32 reference is,
  * the full gradients of the heads / FiLM layers, BN buffer norms after the train-mode forward.
Imports the reference exactly as tests/golden/gen_golden.py does (in-memory stand-ins for its unused, uninstalled imports).
"""
from __future__ import annotations

import json
import os
import sys
import time
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

from hrseg_amd.utils import synth  # noqa: E402
from hrseg_amd.utils.config import hrnet_w48_config  # noqa: E402
from hrseg_amd.utils.hierarchy import get_classes  # noqa: E402

REF = G.REF
LATTICE = 5
CASES_620 = {
    # name: (model, tree file, size, batch, seed)
    "hrnet_hier_tl_620_b2": ("hrnet", "class_tree_tl.json", 620, 2, 620),
    "unet_hier_tl_620_b2": ("unet", "class_tree_tl.json", 620, 2, 621),
}


def projection_vector(name, numel):
    """seeded +-1 vector of a parameter (flattened in the logical [Cout,Cin,kh,kw] order)"""
    return (synth._rng("proj::" + name).integers(0, 2, size=numel).astype(np.float64) * 2.0 - 1.0)


def checksum(a):
    return np.int64(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def forward_loss(model, kind, x, target, num_classes, weights, ref_losses):
    probs, logits = model(x, type=1) if kind == "unet" else model(x)
    targets, s = [], 0
    for n in num_classes:
        targets.append(target[:, s:s + n])
        s += n
    onehots = []
    for z, t in zip(logits, targets):
        oh = torch.nn.functional.one_hot(torch.argmax(torch.softmax(z, 1), 1), z.shape[1]).permute(0, 3, 1, 2).to(z.dtype)
        onehots.append(torch.where(t == -1, 0, oh))
    loss, ces, dices = 0.0, [], []
    for L, (z, t) in enumerate(zip(logits, targets)):
        ce = ref_losses.CrossEntropyLoss()(z, t, class_weight=weights[L], logits_input=True)
        dice = ref_losses.SoftDiceLoss()(z, t, class_weight=weights[L], logits_input=True)
        ces.append(ce)
        dices.append(dice)
        loss = loss + ce + (dice if dice is not None else 0.0)
    cons = ref_losses.hierarchical_consistency_loss(onehots, model.levels, model.parent_of)
    loss = loss + cons
    return dict(probs=probs, logits=logits, targets=targets, onehots=onehots, ces=ces, dices=dices, cons=cons, loss=loss)


def grad_stats(model):
    names, norms, projs = [], [], []
    for n, p in model.named_parameters():
        g = p.grad.detach().double().reshape(-1).numpy()
        names.append(n)
        norms.append(float(np.sqrt((g * g).sum())))
        projs.append(float((g * projection_vector(n, g.size)).sum()))
    return names, np.array(norms), np.array(projs)


def make_model(ref_models, kind, tree, size):
    if kind == "unet":
        m = ref_models.UNet(size=size, n_channels=3, hierarchy=tree, model_type=1)
    else:
        m = ref_models.HighResolutionNet(hrnet_w48_config(), hierarchy=tree, model_type=1)
    return synth.fill_state_dict(m)


def run_case(ref_models, ref_losses, name):
    kind, tree_file, size, batch, seed = CASES_620[name]
    tree = json.load(open(os.path.join(REF, tree_file)))
    num_classes = get_classes(tree, full=True)
    weights = G.level_weights_for(tree_file, True)
    x_np, t_np = synth.synthetic_batch(tree, batch, size, seed=seed, hierarchical=True)
    x, target = torch.from_numpy(x_np), torch.from_numpy(t_np)
    out = {"seed": np.int64(seed), "batch": np.int64(batch), "size": np.int64(size), "lattice": np.int64(LATTICE),
           "x_crc32": checksum(x_np), "target_crc32": checksum(t_np), "num_classes": np.array(num_classes)}
    sl = (slice(None), slice(None), slice(0, None, LATTICE), slice(0, None, LATTICE))
    t0 = time.time()
    model = make_model(ref_models, kind, tree, size)
    model.eval()
    with torch.no_grad():
        probs, logits = model(x, type=1) if kind == "unet" else model(x)
    for L, z in enumerate(logits):
        out[f"eval_logits{L}"] = z[sl].numpy().copy()
    model.train()
    r = forward_loss(model, kind, x, target, num_classes, weights, ref_losses)
    for L in range(len(num_classes)):
        z, t, oh = r["logits"][L].detach(), r["targets"][L], r["onehots"][L]
        out[f"logits{L}"] = z[sl].numpy().copy()
        out[f"probs{L}"] = r["probs"][L].detach()[sl].numpy().copy()
        out[f"ce{L}"] = np.float32(r["ces"][L].item())
        out[f"dice{L}"] = np.float32(r["dices"][L].item()) if r["dices"][L] is not None else np.float32(np.nan)
        am = torch.argmax(z, 1)
        out[f"argmax_hist{L}"] = np.bincount(am.reshape(-1).numpy(), minlength=z.shape[1]).astype(np.int64)
        # masked one-hot of the prediction prep: per class, pixels predicted / pixels that are also target == 1
        out[f"onehot_count{L}"] = oh.sum((0, 2, 3)).numpy().astype(np.int64)
        out[f"onehot_hit{L}"] = (oh * (t == 1)).sum((0, 2, 3)).numpy().astype(np.int64)
    out["cons_onehot"] = np.float32(float(r["cons"]))
    out["loss"] = np.float32(r["loss"].item())
    r["loss"].backward()
    names, norms, projs = grad_stats(model)
    out["grad_names"], out["grad_norms"], out["grad_projs"] = np.array(names), norms, projs
    for n, p in model.named_parameters():
        if n.split(".")[0] in ("heads", "films", "classifiers"):
            out["grad::" + n] = p.grad.numpy().copy()
    out["buf_names"] = np.array([n for n, _ in model.named_buffers()])
    out["buf_norms"] = np.array([float(b.double().norm()) for _, b in model.named_buffers()], dtype=np.float64)
    print(f"{name}: fp32 reference done in {time.time() - t0:.0f} s, loss {out['loss']:.6f}", flush=True)
    del model, r

    # the same train-mode evaluation in fp64: the yardstick for the gradients
    t0 = time.time()
    m64 = make_model(ref_models, kind, tree, size).double()
    m64.train()
    r64 = forward_loss(m64, kind, x.double(), target.double(), num_classes, weights, ref_losses)
    out["loss64"] = np.float64(r64["loss"].item())
    for L in range(len(num_classes)):
        out[f"logits64_{L}"] = r64["logits"][L].detach()[sl].numpy().astype(np.float32)
    r64["loss"].backward()
    n64, norms64, projs64 = grad_stats(m64)
    assert n64 == names
    out["grad_norms64"], out["grad_projs64"] = norms64, projs64
    print(f"{name}: fp64 reference done in {time.time() - t0:.0f} s, loss {out['loss64']:.9f}", flush=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {os.path.getsize(os.path.join(HERE, name + '.npz')) / 2**20:.2f} MiB")


def main():
    ref_models, ref_losses = G.import_reference()
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or list(CASES_620)):
        run_case(ref_models, ref_losses, name)


if __name__ == "__main__":
    main()
