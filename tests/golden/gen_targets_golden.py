"""Golden vectors for the label-PNG -> ternary-target encoding, produced by the REFERENCE's own
SegDataset methods (Data/dataset.py: separate_masks / traverse_tree / process_ignore_values and the
mask post-processing of __getitem__), run in the build container only:

    python tests/golden/gen_targets_golden.py        # writes tests/golden/targets.npz

Data/dataset.py's module header imports skimage.io and torchvision.transforms.functional, which are
not installed offline and are not used by the encoding methods; empty in-memory stand-ins satisfy the
import statements (same approach as gen_golden.py / SURVEY.md section 8c).  No file I/O, transforms or
augmentation are involved: the label images are synthetic pixel-value maps built here.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_dataset():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub("skimage")
    stub("skimage.io", imread=None)
    stub("torchvision")
    stub("torchvision.transforms")
    stub("torchvision.transforms.functional")
    sys.path.insert(0, REF)
    from Data import dataset as ref_dataset
    return ref_dataset


def encode_with_reference(ds, label):
    """label [H,W] uint8 -> [C,H,W] fp32, following SegDataset.__getitem__ (dataset.py:397-470) with an
    identity spatial transform: ToTensor-style /255 scaling, the 0/255 re-binarisation, stack / 255,
    process_ignore_values for model_type 1."""
    y = ds.separate_masks(label)
    y = [torch.from_numpy(m.astype(np.uint8))[None].float() / 255.0 for m in y]
    y = [torch.where(t < 0.5, 0, 255) for t in y]
    y = torch.stack(y, dim=0) / 255.0
    if ds.model_type == 1:
        name_to_index = {row["class_name"]: idx for idx, row in ds.class_map.iterrows()}
        y = ds.process_ignore_values(ds.class_tree, y, name_to_index)
    return y.permute(1, 0, 2, 3)[0].numpy().astype(np.float32)


def main():
    ref = import_dataset()
    g = np.random.Generator(np.random.PCG64(11))
    out = {}
    for tag, tree_file, map_file in (("tl", "class_tree_tl.json", "class_map.csv"),
                                     ("ext", "class_tree_tl_extended.json", "class_map_extended.csv")):
        tree = json.load(open(os.path.join(REF, tree_file)))
        cmap = pd.read_csv(os.path.join(REF, map_file))
        vals = [int(v) for v in cmap["pixel_val"] if pd.notna(v)]     # parents have no pixel value ("None" -> NaN)
        # every mapped pixel value, plus two values no class owns (unlabelled pixels)
        palette = np.array(vals + [7, 200], dtype=np.uint8)
        labels = palette[g.integers(0, len(palette), size=(3, 20, 28))]
        labels[2, :10] = 0                     # a stretch of pure background
        out[f"label_{tag}"] = labels
        for model_type, kind in ((1, "hier"), (0, "flat")):
            ds = ref.SegDataset([], [], clss_t=tree, clss_m=cmap, classes=len(cmap), model_type=model_type)
            out[f"target_{tag}_{kind}"] = np.stack([encode_with_reference(ds, lab) for lab in labels])
    np.savez_compressed(os.path.join(HERE, "targets.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype, np.unique(v)[:8])


if __name__ == "__main__":
    main()
