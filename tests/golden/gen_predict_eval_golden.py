"""Golden vectors for the flat-model parent synthesis of the reference's predictEval.py (get_parent_masks :85-129,
combine_levels :134-185), produced by the REFERENCE's own functions, run in the build container only:

    python tests/golden/gen_predict_eval_golden.py        # writes tests/golden/predict_eval.npz

predictEval.py's module header imports cv2, matplotlib, skimage.io, the data loaders, train.py and the yacs
config -- none installed offline, none used by the two functions; empty in-memory stand-ins satisfy the import
statements (same approach as gen_golden.py / gen_targets_golden.py).  Inputs are synthetic
one-hot leaf predictions / targets built here.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
DATA = os.path.join(os.path.dirname(os.path.dirname(HERE)), "restrictive-hierarchical-semantic-segmentation_amd", "data")


def import_predict_eval():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub("cv2")
    stub("matplotlib")
    stub("matplotlib.pyplot")
    stub("skimage")
    stub("skimage.io", imread=None)
    data = stub("Data")
    data.dataloaders = stub("Data.dataloaders")
    models_pkg = stub("Models")
    models_pkg.models = stub("Models.models")
    metrics = stub("Metrics")
    metrics.performance_metrics = stub("Metrics.performance_metrics")
    stub("train", get_classes=None, get_metrics=None)
    stub("config", config=None, update_config=None)
    sys.path.insert(0, REF)
    import predictEval
    return predictEval


def main():
    pe = import_predict_eval()
    out = {}
    for tree_file in ("class_tree_tl.json", "class_tree_tl_extended.json"):
        tree = json.load(open(os.path.join(DATA, tree_file)))
        tag = "tl" if tree_file == "class_tree_tl.json" else "ext"
        ch = pe.children_map(tree)
        order = pe.bfs_order(tree)
        leaves = [n for n in order if not ch.get(n)]
        parents = [n for n in order if ch.get(n)]
        name_to_index = {n: i for i, n in enumerate(leaves)}
        g = np.random.Generator(np.random.PCG64(11 + len(leaves)))
        B, H, W = 2, 9, 13
        lab_p = g.integers(0, len(leaves), (B, H, W))
        lab_t = g.integers(0, len(leaves), (B, H, W))
        X = torch.from_numpy(np.moveaxis(np.eye(len(leaves), dtype=np.float32)[lab_p], -1, 1).copy())
        Y = torch.from_numpy(np.moveaxis(np.eye(len(leaves), dtype=np.float32)[lab_t], -1, 1).copy())
        Y[0, :, 0, :3] = 0.0                      # a few unlabeled pixels: every leaf channel 0
        par_x, par_y, names = pe.get_parent_masks([X], [Y], tree, name_to_index)
        assert names == parents
        lv_x = pe.combine_levels([X], par_x, tree, leaves, parents)
        lv_y = pe.combine_levels([Y], par_y, tree, leaves, parents)
        out[f"{tag}_X"], out[f"{tag}_Y"] = X.numpy(), Y.numpy()
        out[f"{tag}_parents_X"], out[f"{tag}_parents_Y"] = par_x[0].numpy(), par_y[0].numpy()
        out[f"{tag}_nlevels"] = np.array(len(lv_x))
        for L, (a, b) in enumerate(zip(lv_x, lv_y)):
            out[f"{tag}_level{L}_X"], out[f"{tag}_level{L}_Y"] = a.numpy(), b.numpy()
        out[f"{tag}_levels_bfs"] = np.array(json.dumps(pe.levels_bfs(tree)))
        out[f"{tag}_parent_names"] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(HERE, "predict_eval.npz"), **out)
    print("wrote predict_eval.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
