"""Evaluation-side level synthesis for flat (model_type 0) models on the MI355X path.

Call surface of the reference's predictEval.py helpers (:36-185): ``children_map``, ``bfs_order``, ``levels_bfs``,
``descendant_leaves``, ``get_parent_masks``, ``combine_levels``.  A flat model predicts leaf classes only; to
score it per hierarchy level the reference synthesises every parent channel as the union of its descendant
leaves ("any > 0") and stitches per-level tensors from leaf and parent channels.  Both steps are one HIP kernel
here (hrseg_combine_levels: per output channel a bit mask over the input channels and a copy/union flag); the
tree walks stay host Python.  Image / CSV writers and the CLI of predictEval.py are out of scope.
"""
from __future__ import annotations

from collections import deque

import torch

from . import ops


def children_map(tree):
    """node -> list of direct children (empty for leaves), every node of the nested dict (predictEval.py:36-47)"""
    ch, stack = {}, [tree]
    while stack:
        t = stack.pop()
        for k, v in t.items():
            if isinstance(v, dict) and len(v) > 0:
                ch[k] = list(v.keys())
                stack.append(v)
            else:
                ch[k] = []
    return ch


def bfs_order(tree):
    """node names breadth first (predictEval.py:49-58)"""
    q, order = deque(tree.items()), []
    while q:
        name, sub = q.popleft()
        order.append(name)
        if isinstance(sub, dict) and len(sub) > 0:
            q.extend(sub.items())
    return order


def levels_bfs(tree):
    """names per depth, breadth first (predictEval.py:61-72)"""
    levels, q = [], deque((n, s, 0) for n, s in tree.items())
    while q:
        name, sub, d = q.popleft()
        if len(levels) <= d:
            levels.append([])
        levels[d].append(name)
        if isinstance(sub, dict) and len(sub) > 0:
            q.extend((cn, cs, d + 1) for cn, cs in sub.items())
    return levels


def descendant_leaves(node, children, is_leaf):
    if is_leaf[node]:
        return [node]
    out = []
    for c in children[node]:
        out.extend(descendant_leaves(c, children, is_leaf))
    return out


def get_parent_masks(in_out, target, tree, leaf_index):
    """([X], [Y], tree, {leaf name: channel}) -> ([parents of X], [parents of Y], parent names in BFS order);
    X, Y are [B, n_leaves, H, W]; a parent channel is 1 where any of its descendant leaves is > 0
    (predictEval.py:85-129, same validation errors)."""
    X, Y = in_out[0], target[0]
    C = X.shape[1]
    children = children_map(tree)
    names = bfs_order(tree)
    is_leaf = {n: len(children[n]) == 0 for n in names}
    parent_names = [n for n in names if not is_leaf[n]]
    masks = []
    for p in parent_names:
        leaves = descendant_leaves(p, children, is_leaf)
        if len(leaves) == 0:
            raise ValueError(f"Parent '{p}' has no descendant leaves.")
        bad = [l for l in leaves if l not in leaf_index]
        if bad:
            raise KeyError(f"Missing leaf_index entries for {bad} (needed by parent '{p}').")
        idxs = [leaf_index[l] for l in leaves]
        if min(idxs) < 0 or max(idxs) >= C:
            raise IndexError(f"Parent '{p}' has leaf indices out of bounds: {idxs} with C={C}.")
        masks.append(sum(1 << i for i in set(idxs)))
    flags = [1] * len(masks)
    out_parents = ops.combine_levels(X, None, masks, flags).to(X.dtype)
    target_parents = ops.combine_levels(Y, None, masks, flags).to(Y.dtype)
    return [out_parents], [target_parents], parent_names


def combine_levels(leaves_list, parents_list, tree: dict, leaf_order=None, parent_order=None):
    """([X_leaves], [X_parents], tree) -> one [B, C_level, H, W] tensor per depth, channels in BFS order, each a
    copy of its leaf or parent channel (predictEval.py:134-185, same KeyErrors)"""
    X_leaves, X_par = leaves_list[0], parents_list[0]
    B, C0, H, W = X_leaves.shape
    levels = levels_bfs(tree)
    children = children_map(tree)
    all_names = [n for lvl in levels for n in lvl]
    is_leaf = {n: len(children.get(n, [])) == 0 for n in all_names}
    leaf_names = [n for n in all_names if is_leaf[n]]
    parent_names = [n for n in all_names if not is_leaf[n]]
    leaf_order = leaf_names if leaf_order is None else leaf_order
    parent_order = parent_names if parent_order is None else parent_order
    leaf_index = {n: i for i, n in enumerate(leaf_order)}
    parent_index = {n: i for i, n in enumerate(parent_order)}
    missing_leaves = [n for n in leaf_names if n not in leaf_index]
    missing_parents = [n for n in parent_names if n not in parent_index]
    if missing_leaves:
        raise KeyError(f"leaf_order is missing leaves: {missing_leaves}")
    if missing_parents:
        raise KeyError(f"parent_order is missing parents: {missing_parents}")
    out = []
    for lvl in levels:
        if len(lvl) == 0:
            out.append(torch.zeros((B, 0, H, W), device=X_leaves.device, dtype=X_leaves.dtype))
            continue
        masks = [1 << (leaf_index[n] if is_leaf[n] else C0 + parent_index[n]) for n in lvl]
        out.append(ops.combine_levels(X_leaves, X_par, masks, [0] * len(masks)).to(X_leaves.dtype))
    return out
