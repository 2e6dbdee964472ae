"""The oracle (CPU restatement) against golden vectors generated from the reference
itself (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle import models as OM
from oracle import train_step as OT
from tests.helpers import CASES, build_model, level_weights_for, load_golden, load_tree, rel_err

TOL = 2e-5


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference(name):
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    torch.manual_seed(0)
    model = build_model(OM, kind, hier, tree, size)
    x, target = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    num_classes = [int(v) for v in g["num_classes"]]
    weights = level_weights_for(tree_file, hier)

    # same state_dict keys as the reference (names stored with the fixture)
    names = [n for n, _ in model.named_parameters()]
    assert names == list(g["grad_names"])
    assert [n for n, _ in model.named_buffers()] == list(g["buf_names"])

    model.eval()
    with torch.no_grad():
        _, logits = model(x, type=1 if hier else 0) if kind == "unet" else model(x)
    logits = logits if hier else [logits]
    for L, z in enumerate(logits):
        assert rel_err(z.numpy(), g[f"eval_logits{L}"]) < TOL

    model.train()
    out = OT.forward_loss(model, x, target, num_classes, weights, hierarchical=hier, is_unet=(kind == "unet"))
    for L, z in enumerate(out["logits"]):
        assert rel_err(z.detach().numpy(), g[f"logits{L}"]) < TOL
        ce, dice = out["parts"][L]
        assert abs(ce.item() - g[f"ce{L}"]) < TOL * max(1, abs(g[f"ce{L}"]))
        assert abs(dice.item() - g[f"dice{L}"]) < TOL * max(1, abs(g[f"dice{L}"]))
    if hier:
        for L, p in enumerate(out["probs"]):
            assert rel_err(p.detach().numpy(), g[f"probs{L}"]) < TOL
        assert abs(out["cons"].item() - g["cons_onehot"]) < 1e-6
        cons_p = OL.hierarchical_consistency_loss([p.detach() for p in out["probs"]], model.levels, model.parent_of)
        assert abs(float(cons_p) - g["cons_probs"]) < 1e-6
    assert abs(out["loss"].item() - g["loss"]) < TOL * abs(g["loss"])

    out["loss"].backward()
    norms = np.array([0.0 if p.grad is None else float(p.grad.double().norm()) for _, p in model.named_parameters()])
    ref = g["grad_norms"]
    scale = np.maximum(ref, 1e-3 * ref.max())
    assert np.max(np.abs(norms - ref) / scale) < 2e-3
    for key in g.files:
        if key.startswith("grad::"):
            p = dict(model.named_parameters())[key[6:]]
            # conv biases in front of a BN have an exactly-zero true gradient: absolute floor
            assert np.abs(p.grad.numpy() - g[key]).max() < 2e-3 * np.abs(g[key]).max() + 1e-7, key
    bufs = np.array([float(b.double().norm()) for _, b in model.named_buffers()])
    assert np.max(np.abs(bufs - g["buf_norms"]) / np.maximum(g["buf_norms"], 1e-6)) < 1e-4


def test_loss_edge_cases():
    g = load_golden("loss_cases")
    w = [float(v) for v in g["w"]]
    z = torch.from_numpy(g["z"]).requires_grad_(True)
    t = torch.from_numpy(g["t"])
    ce = OL.cross_entropy_loss(z, t, True, w)
    dice = OL.soft_dice_loss(z, t, True, w)
    assert abs(ce.item() - g["ce"]) < 1e-6 and abs(dice.item() - g["dice"]) < 1e-6
    (ce + dice).backward()
    assert rel_err(z.grad.numpy(), g["dz"]) < 1e-5
    ce2 = OL.cross_entropy_loss(torch.from_numpy(g["z_all"]), torch.from_numpy(g["t_all"]), True, w)
    assert abs(ce2.item() - g["ce_all"]) < 1e-6
    assert OL.soft_dice_loss(torch.from_numpy(g["z_all"]), torch.from_numpy(g["t_all"]), True, w) is None
    assert bool(g["dice_all_is_none"])
