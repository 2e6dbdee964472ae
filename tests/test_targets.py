"""Label image -> target planes: oracle vs the reference's golden vectors on CPU,
the HIP encoder vs both on the GPU."""
import csv
import json
import os

import numpy as np
import pytest
import torch

from oracle import targets as OT

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "restrictive-hierarchical-semantic-segmentation_amd", "data")
TREES = {"tl": ("class_tree_tl.json", "class_map.csv"), "ext": ("class_tree_tl_extended.json", "class_map_extended.csv")}


def load(tag):
    tree_file, map_file = TREES[tag]
    tree = json.load(open(os.path.join(DATA, tree_file)))
    rows = list(csv.DictReader(open(os.path.join(DATA, map_file))))
    name2pix = {r["class_name"]: int(r["pixel_val"]) for r in rows if r["pixel_val"] != "None"}
    return tree, rows, name2pix


GOLD = np.load(os.path.join(HERE, "golden", "targets.npz"))


@pytest.mark.parametrize("tag", ["tl", "ext"])
@pytest.mark.parametrize("model_type", [1, 0])
def test_oracle_targets_match_reference(tag, model_type):
    tree, _, name2pix = load(tag)
    want = GOLD[f"target_{tag}_{'hier' if model_type else 'flat'}"]
    got = OT.encode(GOLD[f"label_{tag}"], tree, name2pix, model_type)
    assert got.dtype == np.float32 and np.array_equal(got, want)          # values are {1,0,-1}: bit-exact


def test_oracle_targets_missing_class():
    tree, _, name2pix = load("tl")
    del name2pix["enamel"]
    with pytest.raises(KeyError):
        OT.encode(GOLD["label_tl"], tree, name2pix, 1)


def test_synthetic_targets_agree_with_encoder_semantics():
    """utils.synth.encode_targets (leaf-id label maps, used by bench and the model tests) is the same
    encoding as the reference's pixel-value path"""
    from hrseg_amd.utils import synth
    from hrseg_amd.utils.hierarchy import level_order_names, _find
    tree, _, name2pix = load("ext")
    leaves = [n for n in level_order_names(tree) if not _find(tree, n)]
    g = np.random.Generator(np.random.PCG64(3))
    lab = g.integers(0, len(leaves), size=(2, 9, 13))
    pix = np.array([name2pix[n] for n in leaves], dtype=np.uint8)[lab]
    for hier in (True, False):
        assert np.array_equal(synth.encode_targets(lab, tree, hier), OT.encode(pix, tree, name2pix, 1 if hier else 0))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tl", "ext"])
@pytest.mark.parametrize("model_type", [1, 0])
def test_gpu_encoder_matches_golden(tag, model_type):
    from hrseg_amd.Data import TargetEncoder
    tree, rows, _ = load(tag)
    enc = TargetEncoder(tree, rows, model_type)
    got = enc(torch.from_numpy(GOLD[f"label_{tag}"]).cuda())
    want = GOLD[f"target_{tag}_{'hier' if model_type else 'flat'}"]
    assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_gpu_encoder_full_size_and_2d_input():
    """BASELINE size (4 x 620 x 620) against the oracle, every uint8 value present; [H,W] input"""
    from hrseg_amd.Data import TargetEncoder
    tree, rows, name2pix = load("tl")
    g = np.random.Generator(np.random.PCG64(5))
    lab = g.integers(0, 256, size=(4, 620, 620)).astype(np.uint8)
    enc = TargetEncoder(tree, {r["class_name"]: r["pixel_val"] for r in rows}, 1)
    got = enc(torch.from_numpy(lab).cuda()).cpu().numpy()
    assert np.array_equal(got, OT.encode(lab, tree, name2pix, 1))
    one = enc(torch.from_numpy(lab[0]).cuda()).cpu().numpy()
    assert one.shape == (1, 8, 620, 620) and np.array_equal(one[0], got[0])
    # size-independent property: per level, inside the parent's area the children are one-hot or all 0
    t = torch.from_numpy(got)
    child = t[:, 4:8]
    inside = (t[:, 3:4] == 1)
    assert bool(((child == -1).all(1, keepdim=True) == ~inside).all())
    assert bool(((child == 1).sum(1, keepdim=True)[inside] == 1).all())


@pytest.mark.gpu
def test_gpu_encoder_errors():
    from hrseg_amd.Data import TargetEncoder
    tree, rows, _ = load("tl")
    with pytest.raises(KeyError):
        TargetEncoder(tree, [r for r in rows if r["class_name"] != "pulp"], 1)
    enc = TargetEncoder(tree, rows, 1)
    with pytest.raises(TypeError):
        enc(torch.zeros(2, 4, 4, dtype=torch.int64).cuda())
