"""Evaluation-side level synthesis for flat (model_type 0) models on the MI355X path.

Call surface of the reference's predictEval.py helpers (:36-185): ``children_map``, ``bfs_order``, ``levels_bfs``,
``descendant_leaves``, ``get_parent_masks``, ``combine_levels``.  A flat model predicts leaf classes only; to
score it per hierarchy level the reference synthesises every parent channel as the union of its descendant
leaves ("any > 0") and stitches per-level tensors from leaf and parent channels.  Both steps are one HIP kernel
here (hrseg_combine_levels: per output channel a bit mask over the input channels and a copy/union flag); the
tree walks run over one shared index (utils/hierarchy.TreeIndex).  The per-batch body of predict() (:305-573) --
prediction prep, metrics, the PNG dump and metrics.csv -- is below; only the CLI / fold / dataset plumbing is out of scope.
"""
from __future__ import annotations

import torch

from . import ops
from .utils.hierarchy import TreeIndex


def children_map(tree):
    """node -> list of direct children (empty for leaves), for every node of the nested dict (predictEval.py:36-47)"""
    return {name: list(kids) for name, kids in TreeIndex(tree).children.items()}


def bfs_order(tree):
    """node names breadth first (predictEval.py:49-58)"""
    return list(TreeIndex(tree).order)


def levels_bfs(tree):
    """names per depth, breadth first (predictEval.py:61-72)"""
    return [list(lvl) for lvl in TreeIndex(tree).levels]


def descendant_leaves(node, children, is_leaf):
    """leaves below `node` given a children map and a leaf predicate map (predictEval.py:74-83), depth first"""
    found, todo = [], [node]
    while todo:
        cur = todo.pop()
        if is_leaf[cur]:
            found.append(cur)
        else:
            todo.extend(reversed(children[cur]))
    return found


def _channel_mask(channels):
    m = 0
    for c in channels:
        m |= 1 << c
    return m


def get_parent_masks(in_out, target, tree, leaf_index):
    """([X], [Y], tree, {leaf name: channel}) -> ([parents of X], [parents of Y], parent names in BFS order);
    X, Y are [B, n_leaves, H, W]; a parent channel is 1 where any of its descendant leaves is > 0
    (predictEval.py:85-129; same exception types and messages)."""
    X, Y = in_out[0], target[0]
    n_ch = X.shape[1]
    index = TreeIndex(tree)
    parents = index.parent_names
    masks = []
    for p in parents:
        below = index.leaves[p]
        if not below:
            raise ValueError(f"Parent '{p}' has no descendant leaves.")
        unknown = [leaf for leaf in below if leaf not in leaf_index]
        if unknown:
            raise KeyError(f"Missing leaf_index entries for {unknown} (needed by parent '{p}').")
        idxs = [leaf_index[leaf] for leaf in below]
        if any(i < 0 or i >= n_ch for i in idxs):
            raise IndexError(f"Parent '{p}' has leaf indices out of bounds: {idxs} with C={n_ch}.")
        masks.append(_channel_mask(idxs))
    union = [1] * len(masks)
    return ([ops.combine_levels(X, None, masks, union).to(X.dtype)], [ops.combine_levels(Y, None, masks, union).to(Y.dtype)],
            parents)


def combine_levels(leaves_list, parents_list, tree: dict, leaf_order=None, parent_order=None):
    """([X_leaves], [X_parents], tree) -> one [B, C_level, H, W] tensor per depth, channels in BFS order, each a
    copy of its leaf or parent channel (predictEval.py:134-185; same KeyErrors)"""
    X_leaves, X_par = leaves_list[0], parents_list[0]
    B, n_leaf_ch, H, W = X_leaves.shape
    index = TreeIndex(tree)
    leaf_order = index.leaf_names if leaf_order is None else leaf_order
    parent_order = index.parent_names if parent_order is None else parent_order
    where = {n: i for i, n in enumerate(leaf_order)}
    where_parent = {n: n_leaf_ch + i for i, n in enumerate(parent_order)}
    absent = [n for n in index.leaf_names if n not in where]
    if absent:
        raise KeyError(f"leaf_order is missing leaves: {absent}")
    absent = [n for n in index.parent_names if n not in where_parent]
    if absent:
        raise KeyError(f"parent_order is missing parents: {absent}")
    where.update(where_parent)                      # input channel of every node: leaves first, parents behind them
    out = []
    for names in index.levels:
        copy_masks = [1 << where[n] for n in names]
        out.append(ops.combine_levels(X_leaves, X_par, copy_masks, [0] * len(copy_masks)).to(X_leaves.dtype))
    return out


# ----------------------------------------------------------------------------- predict(): batch body, loop, writers
def prediction_prep(output_logits, target, args, class_tree):
    """The per-batch prediction prep of the reference's predict() (predictEval.py:336-441) for one batch:
    (model logits, raw target [B, sum C, H, W]) -> (output_class per level, eval_targets per level), both zeroed where
    the level's target is -1.  Hierarchical models: soft-max -> arg-max -> one-hot per level (:426-432, one
    hrseg_predict_metrics launch per level).  Flat models: one-hot over the leaves, parents synthesised as the union of
    their descendant leaves and the per-level tensors stitched in BFS order (:381-386)."""
    from .train import split_targets
    if args.model_type == 0:
        logits = output_logits if torch.is_tensor(output_logits) else output_logits[0]
        # the leaf one-hot is NOT masked here: the reference synthesises the parents from the plain arg-max one-hot
        # (predictEval.py:361-386) and masks every level afterwards (:435-439) -- a parent's target is never -1, so its
        # prediction keeps the pixels whose LEAF target is -1.  (An all-zero target makes the kernel's mask a no-op.)
        onehot, _ = ops.predict_metrics(logits.detach(), ops.zeros(logits.shape, torch.float32, logits.device), child=False,
                                        mask_pred=True)
        index = TreeIndex(class_tree)
        name_to_index = {n: i for i, n in enumerate(index.leaf_names)}
        parent_class, parent_target, _ = get_parent_masks([onehot], [target], class_tree, name_to_index)
        output_class = combine_levels([onehot], parent_class, class_tree, index.leaf_names, index.parent_names)
        targets = combine_levels([target], parent_target, class_tree, index.leaf_names, index.parent_names)
        output_class = [torch.where(t == -1, torch.zeros_like(o), o) for o, t in zip(output_class, targets)]
    else:
        targets = split_targets(target, args)
        output_class = [ops.predict_metrics(z.detach(), t, child=(L > 0), mask_pred=True)[0]
                        for L, (z, t) in enumerate(zip(output_logits, targets))]
    eval_targets = [t.clamp_min(0.0) for t in targets]              # -1 -> 0 (:435-439)
    return output_class, eval_targets


def write_png_gray(path, mask_u8):
    """8-bit grayscale PNG of a [H, W] uint8 array (what the reference writes with cv2.imwrite, predictEval.py:512;
    cv2 is not a dependency here: the format is 30 lines of zlib + struct)"""
    import struct
    import zlib
    import numpy as np
    a = np.ascontiguousarray(mask_u8, dtype=np.uint8)
    h, w = a.shape
    raw = b"".join(b"\x00" + a[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def save_prediction_images(output_class, save_dir, filename):
    """first image of the batch, one 0/255 PNG per class under <save_dir>/<class index>/<filename> (predictEval.py:500-513)"""
    import os
    k = 0
    for level in output_class:
        planes = (level[0] > 0.5).to(torch.uint8).mul_(255).cpu().numpy()
        for plane in planes:
            os.makedirs(os.path.join(save_dir, str(k)), exist_ok=True)
            write_png_gray(os.path.join(save_dir, str(k), filename), plane)
            k += 1


def write_metrics_csv(path, accuracy, iou, dice, precision, recall, class_metrics):
    """metrics.csv of predict() (predictEval.py:556-573): one "Average" row, one row per class"""
    import csv
    import numpy as np
    with open(path, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["Type", "Class", "Accuracy", "IoU", "Dice", "Precision", "Recall"])
        wr.writerow(["Average", "All"] + [float(np.mean(v)) for v in (accuracy, iou, dice, precision, recall)])
        for c, m in enumerate(class_metrics):
            wr.writerow(["Class", c] + [float(np.mean(np.asarray(m[k]))) for k in ("accuracy", "iou", "dice", "precision", "recall")])


@torch.no_grad()
def predict_loop(model, device, test_loader, args, class_tree, Accuracy, Iou, perf_measure, Precision, Recall,
                 save_dir=None, target_paths=None):
    """The per-fold body of the reference's predict() (predictEval.py:305-573) on a built model and loader: eval-mode
    forward, prediction_prep, get_metrics per batch, optional PNG dump of each batch's first image and metrics.csv.
    -> dict(accuracy, iou, dice, precision, recall, class_metrics, performance)."""
    import os
    import numpy as np
    from . import train as T
    model.eval()
    n_cls = sum(args.num_classes_full) if hasattr(args, "num_classes_full") else sum(args.num_classes)
    acc2, iou2, dice2, prec2, rec2, perf = [], [], [], [], [], []
    cls2 = T._new_class_metrics(n_cls)
    for i, (data, target) in enumerate(test_loader):
        data, target = data.to(device), target.to(device)
        _, output_logits = T._model_call(model, data, args, class_tree)
        output_class, eval_targets = prediction_prep(output_logits, target, args, class_tree)
        cls2, acc2, iou2, dice2, prec2, rec2, no_bg = T.get_metrics(output_class, eval_targets, acc2, iou2, dice2, prec2, rec2,
                                                                    Accuracy, Iou, perf_measure, Precision, Recall, device,
                                                                    cls2, args)
        perf.append(float(no_bg.mean()))
        if save_dir is not None:
            name = os.path.basename(target_paths[i]) if target_paths is not None else f"{i:05d}.png"
            save_prediction_images(output_class, save_dir, name)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        write_metrics_csv(os.path.join(save_dir, "metrics.csv"), acc2, iou2, dice2, prec2, rec2, cls2)
    return dict(accuracy=float(np.mean(acc2)), iou=float(np.mean(iou2)), dice=float(np.mean(dice2)),
                precision=float(np.mean(prec2)), recall=float(np.mean(rec2)), class_metrics=cls2, performance=perf)
