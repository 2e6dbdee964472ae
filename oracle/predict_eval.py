"""ORACLE (test infrastructure, not product): flat-model parent synthesis on CPU (numpy).

Restates /root/reference/predictEval.py:36-185 (children_map, bfs_order, levels_bfs, descendant_leaves,
get_parent_masks, combine_levels).  Pinned by tests/golden/predict_eval.npz, which
tests/golden/gen_predict_eval_golden.py produces with the reference's own functions.
"""
from __future__ import annotations

from collections import deque

import numpy as np


def children_map(tree):                                   # predictEval.py:36-47
    ch, stack = {}, [tree]
    while stack:
        t = stack.pop()
        for k, v in t.items():
            ch[k] = list(v.keys()) if isinstance(v, dict) and v else []
            if ch[k]:
                stack.append(v)
    return ch


def bfs_order(tree):                                      # predictEval.py:49-58
    q, order = deque(tree.items()), []
    while q:
        n, sub = q.popleft()
        order.append(n)
        if isinstance(sub, dict) and sub:
            q.extend(sub.items())
    return order


def levels_bfs(tree):                                     # predictEval.py:61-72
    levels, q = [], deque((n, s, 0) for n, s in tree.items())
    while q:
        n, sub, d = q.popleft()
        while len(levels) <= d:
            levels.append([])
        levels[d].append(n)
        if isinstance(sub, dict) and sub:
            q.extend((cn, cs, d + 1) for cn, cs in sub.items())
    return levels


def _leaves_under(node, ch):
    return [node] if not ch[node] else [l for c in ch[node] for l in _leaves_under(c, ch)]


def get_parent_masks(X, Y, tree, leaf_index):             # predictEval.py:85-129
    ch = children_map(tree)
    parents = [n for n in bfs_order(tree) if ch[n]]
    def union(A):
        return np.stack([(A[:, sorted({leaf_index[l] for l in _leaves_under(p, ch)})] > 0).any(1) for p in parents], 1).astype(A.dtype)
    return union(X), union(Y), parents


def combine_levels(X_leaves, X_par, tree, leaf_order, parent_order):      # predictEval.py:134-185
    ch = children_map(tree)
    li = {n: i for i, n in enumerate(leaf_order)}
    pi = {n: i for i, n in enumerate(parent_order)}
    return [np.stack([X_par[:, pi[n]] if ch[n] else X_leaves[:, li[n]] for n in lvl], 1) for lvl in levels_bfs(tree)]
