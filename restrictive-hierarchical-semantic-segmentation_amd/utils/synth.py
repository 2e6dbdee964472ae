"""Deterministic synthetic inputs and name-keyed weights (numpy only).

No dataset or checkpoint is reachable offline, so every test, fixture and
benchmark draws from these recipes.  They depend only on integer seeds and
numpy's PCG64 stream, hence give identical bits on every host: the golden
fixtures under tests/golden/ never store weights, both sides rebuild them
from the parameter *names*.

Target encoding follows the reference's dataloader contract
(Data/dataset.py:227-265): root channels are {0,1}; a non-root channel is 1 on
the class, 0 inside its direct parent but off the class, -1 outside the parent.
"""
from __future__ import annotations

import zlib

import numpy as np

from .hierarchy import level_order_names, _find

README_LEVEL_WEIGHTS_TL = [[0.0297, 1.577, 0.9619, 0.1770], [1.5432, 0.2638, 1.0413, 3.9722]]
README_LEVEL_WEIGHTS_FLAT = [[0.0285, 1.5159, 0.9227, 1.4842, 0.2532, 1.0, 3.8021]]


def _rng(name: str, salt: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) << 8) ^ salt))


def tensor_for(name: str, shape, salt: int = 0) -> np.ndarray:
    """Value of state_dict entry ``name`` (fp32, or int64 for the BN counter)."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    u = _rng(name, salt).random(shape)            # float64 in [0,1)
    if leaf == "running_mean":
        v = (u - 0.5) * 0.2
    elif leaf == "running_var":
        v = 0.5 + u
    elif leaf == "weight" and len(shape) == 1:    # BN gamma
        v = 0.5 + u
    elif leaf == "bias":
        v = (u - 0.5) * 0.2
    elif leaf == "weight":                        # conv [O,I,kh,kw] / linear [O,I]
        fan_in = int(np.prod(shape[1:]))
        v = (u - 0.5) * 2.0 * np.sqrt(3.0 / fan_in)   # var = 1/fan_in
    else:
        v = (u - 0.5) * 0.2
    return v.astype(np.float32)


def fill_state_dict(module, salt: int = 0):
    """Overwrite every parameter/buffer of a torch module from its name."""
    import torch
    with torch.no_grad():
        for name, t in module.state_dict().items():
            src = torch.from_numpy(tensor_for(name, t.shape, salt))
            t.copy_(src.to(device=t.device, dtype=t.dtype))
    return module


def make_image(batch: int, size: int, seed: int = 0, channels: int = 3) -> np.ndarray:
    """[B,C,S,S] fp32 in [-1,1] (the reference normalises to that range,
    Data/dataloaders.py:55,63)."""
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    return (g.random((batch, channels, size, size)) * 2.0 - 1.0).astype(np.float32)


def make_label_map(batch: int, size: int, n_leaves: int, seed: int = 0, blob: int = 8) -> np.ndarray:
    """[B,S,S] int64 leaf index per pixel, piecewise constant on blob x blob cells."""
    g = np.random.Generator(np.random.PCG64(2000 + seed))
    cells = (size + blob - 1) // blob
    coarse = g.integers(0, n_leaves, size=(batch, cells, cells))
    lab = np.repeat(np.repeat(coarse, blob, axis=1), blob, axis=2)[:, :size, :size]
    return np.ascontiguousarray(lab.astype(np.int64))


def encode_targets(label_map: np.ndarray, tree: dict, hierarchical: bool = True) -> np.ndarray:
    """label map over the tree's leaves (BFS leaf order) -> [B,Ctot,S,S] fp32.

    hierarchical: every node in BFS order, ternary {1,0,-1}; flat: leaves only,
    one-hot {1,0}."""
    names = level_order_names(tree)
    leaves = [n for n in names if not _find(tree, n)]
    leaf_id = {n: i for i, n in enumerate(leaves)}

    def mask_of(name):
        sub = _find(tree, name)
        if not sub:
            return label_map == leaf_id[name]
        m = np.zeros(label_map.shape, dtype=bool)
        for c in sub:
            m |= mask_of(c)
        return m

    if not hierarchical:
        return np.stack([mask_of(n) for n in leaves], axis=1).astype(np.float32)

    parent = {}

    def link(node, p):
        for k, v in node.items():
            parent[k] = p
            if isinstance(v, dict) and v:
                link(v, k)
    link(tree, None)

    chans = []
    for n in names:
        on = mask_of(n)
        if parent[n] is None:
            chans.append(on.astype(np.float32))
        else:
            inside = mask_of(parent[n])
            chans.append(np.where(on, 1.0, np.where(inside, 0.0, -1.0)).astype(np.float32))
    return np.stack(chans, axis=1)


def synthetic_batch(tree: dict, batch: int, size: int, seed: int = 0, hierarchical: bool = True,
                    blob: int = 8):
    """-> (x [B,3,S,S] fp32, target [B,Ctot,S,S] fp32)."""
    names = level_order_names(tree)
    n_leaves = sum(1 for n in names if not _find(tree, n))
    lab = make_label_map(batch, size, n_leaves, seed, blob)
    return make_image(batch, size, seed), encode_targets(lab, tree, hierarchical)
