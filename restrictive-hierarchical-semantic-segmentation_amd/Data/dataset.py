"""Label image -> model targets on the GPU (the input side of the train step).

Mirror of the target preparation in the reference's `SegDataset` (Data/dataset.py):
`separate_masks`/`traverse_tree` (:41-124, per-node masks, parents = OR of their children, emitted
in level order; leaves only when model_type == 0) followed by `process_ignore_values` (:227-265,
ternary {1, 0, -1} relative to the direct parent) -- for label images that are already at the
training resolution.  The reference does this per sample on the CPU inside the DataLoader with one
full-size mask image per node; here the whole batch is ONE kernel (`hrseg_encode_targets`): a
256-entry bit table maps a pixel value to the set of nodes it switches on, 1 byte read and 4*C bytes
written per pixel, so the label map can travel to the device as uint8 (C*4 times less PCIe traffic
than the float target tensor).

Image decoding, resizing and augmentation (dataset.py:397-447, dataloaders.py) stay outside the path.
"""
from __future__ import annotations

import torch

from .. import ops
from .._lib import require_gpu
from ..utils.hierarchy import level_order_names, _find


class TargetEncoder:
    """`enc = TargetEncoder(class_tree, class_map, model_type); target = enc(label_u8)`.

    class_tree: the nested dict of class_tree_*.json; class_map: the reference's class_map.csv as a
    pandas DataFrame, a list of row dicts, or a {class_name: pixel_val} dict (parents have no value);
    model_type 1 = hierarchical ternary targets for every node, 0 = one-hot over the leaves."""

    def __init__(self, class_tree: dict, class_map, model_type: int = 1, device="cuda"):
        require_gpu()
        self.class_tree, self.model_type = class_tree, int(model_type)
        name2pix = self._name2pix(class_map)
        names = level_order_names(class_tree)
        leaf = {n: not _find(class_tree, n) for n in names}
        self.names = names if self.model_type == 1 else [n for n in names if leaf[n]]
        if len(self.names) > 64:
            raise ValueError("TargetEncoder supports at most 64 target channels")
        chan = {n: i for i, n in enumerate(self.names)}

        parent = {}

        def link(node, p):
            for k, v in node.items():
                parent[k] = p
                if isinstance(v, dict) and v:
                    link(v, k)
        link(class_tree, None)
        self.parent = [(-1 if (self.model_type == 0 or parent[n] is None) else chan[parent[n]]) for n in self.names]

        lut = [0] * 256
        for n in names:
            if not leaf[n]:
                continue
            if n not in name2pix:
                raise KeyError(f"Class '{n}' not found in class_map.")      # as dataset.py:62-64
            v = int(name2pix[n])
            if not 0 <= v <= 255:
                raise ValueError(f"pixel value {v} of class '{n}' does not fit a uint8 label image")
            a = n
            while a is not None:                       # the leaf switches on itself and all its ancestors
                if a in chan:
                    lut[v] |= 1 << chan[a]
                a = parent[a]
        # bit patterns as int64 (two's complement for bit 63)
        self.on_lut = torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in lut], dtype=torch.int64,
                                   device=device)

    @staticmethod
    def _name2pix(class_map):
        if isinstance(class_map, dict):
            items = class_map.items()
        elif hasattr(class_map, "iterrows"):
            items = ((row["class_name"], row["pixel_val"]) for _, row in class_map.iterrows())
        else:
            items = ((row["class_name"], row["pixel_val"]) for row in class_map)
        out = {}
        for name, v in items:
            if v is None or (isinstance(v, str) and v.strip().lower() in ("none", "nan", "")):
                continue
            if isinstance(v, float) and v != v:
                continue
            out[name] = int(float(v))
        return out

    def __call__(self, label: torch.Tensor) -> torch.Tensor:
        """label [B,H,W] (or [H,W]) uint8 pixel values -> [B,C,H,W] fp32 targets on the device"""
        if label.dim() == 2:
            label = label[None]
        if label.dtype != torch.uint8:
            raise TypeError("label image must be uint8 pixel values")
        label = label.to(self.on_lut.device, non_blocking=True)
        return ops.encode_targets(label, self.on_lut, self.parent)
