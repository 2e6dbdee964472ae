"""CPU pins for the host logic of the path (no GPU, no reference at run time): the expected values were
obtained from the reference (a1 values for both shipped trees; predict_eval.npz is written by the
reference's own functions through tests/golden/gen_predict_eval_golden.py) or derived by hand from its text
(tree_util.py:16-140, Models/models.py:804-832)."""
import json

import numpy as np
import pytest
import torch

from tests.helpers import load_golden, load_tree


# ------------------------------------------------------------------ a1: tree helpers (models.py:38-54, 82-98; train.py:86-106)
def test_tree_helpers_on_both_shipped_trees():
    from hrseg_amd.utils.hierarchy import build_hierarchy_indices, child_groups, get_classes, get_level_classes
    tl, ext = load_tree("class_tree_tl.json"), load_tree("class_tree_tl_extended.json")
    assert get_classes(tl, full=True) == [4, 4] and get_classes(tl, full=False) == [3, 4]
    assert get_classes(ext, full=True) == [2, 2, 4, 3] and get_classes(ext, full=False) == [1, 0, 3, 3]
    levels, parent_of, children_of = build_hierarchy_indices(tl)
    assert levels == [["background", "upper", "lower", "tooth"], ["pulp", "dentin", "enamel", "composite"]]
    assert parent_of["pulp"] == "tooth" and parent_of["composite"] == "tooth" and parent_of.get("tooth") is None
    assert children_of["tooth"] == ["pulp", "dentin", "enamel", "composite"] and children_of.get("upper", []) == []
    levels, parent_of, children_of = build_hierarchy_indices(ext)
    assert levels == [["background", "tooth+alveolar"], ["alveolar", "tooth"], ["upper", "lower", "composite", "healthy"],
                      ["pulp", "dentin", "enamel"]]
    assert parent_of["healthy"] == "tooth" and parent_of["upper"] == "alveolar" and parent_of["enamel"] == "healthy"
    assert get_level_classes(tl, inc_parent=False) == {0: ["background", "upper", "lower"], 1: ["pulp", "dentin", "enamel", "composite"]}
    assert get_level_classes(ext, inc_parent=False) == {0: ["background"], 1: [], 2: ["upper", "lower", "composite"],
                                                        3: ["pulp", "dentin", "enamel"]}
    groups = child_groups(*build_hierarchy_indices(ext)[::2])
    assert groups == [[("tooth+alveolar", ["alveolar", "tooth"])],
                      [("alveolar", ["upper", "lower"]), ("tooth", ["composite", "healthy"])],
                      [("healthy", ["pulp", "dentin", "enamel"])]]


# ------------------------------------------------------------------ tree_util (tree_util.py:16-140)
def test_tree_util_on_a_tab_indented_file(tmp_path):
    from hrseg_amd import tree_util as TU
    f = tmp_path / "tree.txt"
    f.write_text("background\ntooth+alveolar\n\talveolar\n\t\tupper\n\t\tlower\n\ttooth\n\t\tcomposite\n\t\thealthy\n\t\t\tpulp\n\t\t\tdentin\n")
    root = TU.create_tree_from_textfile(str(f))
    assert root.name == "Universal class" and [c.name for c in root.children] == ["background", "tooth+alveolar"]
    ta = root.children[1]
    assert [c.name for c in ta.children] == ["alveolar", "tooth"]
    assert [c.name for c in ta.children[1].children[1].children] == ["pulp", "dentin"]
    assert TU.add_channels(root, 0) == 6                       # leaves numbered depth first
    assert TU.getLeafClasses(root, []) == [0, 1, 2, 3, 4, 5]
    assert TU.getLeafClasses(ta.children[1], []) == [3, 4, 5]
    assert TU.find_depth(root) == 4
    TU.add_levels(root, TU.find_depth(root))
    assert root.children[0].level == 3 and ta.level == 3 and ta.children[0].level == 2
    TU.update_channels(root, {0: 10, 1: 11, 2: 12, 3: 13, 4: 14, 5: 15})
    assert TU.getLeafClasses(root, []) == [10, 11, 12, 13, 14, 15]
    bad = tmp_path / "bad.txt"
    bad.write_text("a\n\t\tb\n")
    with pytest.raises(RuntimeError):
        TU.create_tree_from_textfile(str(bad))


# ------------------------------------------------------------------ init_weights (models.py:804-832)
def test_init_weights_prefix_and_suffix_matching(tmp_path, capsys):
    from hrseg_amd.Models import models as PM
    from hrseg_amd.utils.config import hrnet_w48_config
    tree = load_tree("class_tree_tl.json")
    model = PM.HighResolutionNet(hrnet_w48_config(), hierarchy=tree, model_type=1)
    own = model.state_dict()
    g = torch.Generator().manual_seed(3)
    ckpt = {}
    exact = "stem.0.weight"                       # stored with a DataParallel prefix
    ckpt["module." + exact] = torch.randn(own[exact].shape, generator=g)
    suffix_target = "layer1.0.conv1.weight"       # stored under a longer name that only ENDS with the model key
    ckpt["backbone." + suffix_target] = torch.randn(own[suffix_target].shape, generator=g)
    wrong_shape = "stem.3.weight"                 # right name, wrong shape: must be skipped
    ckpt["model." + wrong_shape] = torch.randn(7, 3, 3, 3, generator=g)
    path = tmp_path / "pretrained.pth"
    torch.save({"state_dict": ckpt}, path)
    before = {k: v.clone() for k, v in own.items()}
    model.init_weights(str(path), "cpu")
    after = model.state_dict()
    assert torch.equal(after[exact], ckpt["module." + exact])
    assert torch.equal(after[suffix_target], ckpt["backbone." + suffix_target])
    assert torch.equal(after[wrong_shape], before[wrong_shape])
    untouched = [k for k in own if k not in (exact, suffix_target)]
    assert all(torch.equal(after[k], before[k]) for k in untouched)
    out = capsys.readouterr().out
    assert f"Loaded 2 / {len(own)} layers." in out and "Missing" in out


# ------------------------------------------------------------------ predictEval host helpers + oracle vs the reference's vectors
@pytest.mark.parametrize("tag,tree_file", [("tl", "class_tree_tl.json"), ("ext", "class_tree_tl_extended.json")])
def test_predict_eval_oracle_and_host_helpers_match_the_reference(tag, tree_file):
    from oracle import predict_eval as OP
    import importlib
    tree = load_tree(tree_file)
    g = load_golden("predict_eval")
    levels_ref = json.loads(str(g[f"{tag}_levels_bfs"]))
    parents_ref = json.loads(str(g[f"{tag}_parent_names"]))
    assert OP.levels_bfs(tree) == levels_ref
    ch = OP.children_map(tree)
    leaves = [n for n in OP.bfs_order(tree) if not ch[n]]
    li = {n: i for i, n in enumerate(leaves)}
    X, Y = g[f"{tag}_X"], g[f"{tag}_Y"]
    px, py, names = OP.get_parent_masks(X, Y, tree, li)
    assert names == parents_ref
    assert np.array_equal(px, g[f"{tag}_parents_X"]) and np.array_equal(py, g[f"{tag}_parents_Y"])
    for L, (a, b) in enumerate(zip(OP.combine_levels(X, px, tree, leaves, names), OP.combine_levels(Y, py, tree, leaves, names))):
        assert np.array_equal(a, g[f"{tag}_level{L}_X"]) and np.array_equal(b, g[f"{tag}_level{L}_Y"])
    assert int(g[f"{tag}_nlevels"]) == len(levels_ref)
    PE = importlib.import_module("hrseg_amd.predictEval")       # the product's host-side tree walks (no kernel calls)
    assert PE.levels_bfs(tree) == levels_ref and PE.bfs_order(tree) == OP.bfs_order(tree)
    assert PE.children_map(tree) == ch
