"""Class-tree helpers (host-side, pure Python).

The class tree is a nested dict whose insertion order is the channel order
(reference: class_tree_tl.json, class_tree_tl_extended.json).

Mirrors, by behaviour:
  * ``get_level_classes``        -- reference Models/models.py:82-98
  * ``build_hierarchy_indices``  -- reference Models/models.py:38-54
  * ``get_classes``              -- reference train.py:86-106
  * ``child_groups``             -- the group construction inlined in
                                    Models/models.py:229-238 and :636-645
"""
from __future__ import annotations


def _is_branch(v) -> bool:
    return isinstance(v, dict) and len(v) > 0


def get_level_classes(hierarchy, depth=0, result=None, inc_parent=False):
    """{depth: [names]}; with ``inc_parent`` every node is listed at its depth,
    otherwise only leaves are (a depth that holds no leaf still gets an empty
    list, exactly as the reference's traversal creates it)."""
    if result is None:
        result = {}
    if not _is_branch(hierarchy):
        return result
    names = result.setdefault(depth, [])
    for name, sub in hierarchy.items():
        if inc_parent or not sub:
            names.append(name)
        if isinstance(sub, dict):
            get_level_classes(sub, depth + 1, result, inc_parent)
    return result


def build_hierarchy_indices(hierarchy):
    """-> (levels, parent_of, children_of); ``levels[d]`` lists every node at depth d."""
    by_depth = get_level_classes(hierarchy, inc_parent=True)
    levels = [by_depth[d] for d in sorted(by_depth)]
    parent_of, children_of = {}, {}

    stack = [(hierarchy, None)]
    # iterative pre-order walk that keeps insertion order
    while stack:
        node, parent = stack.pop()
        pending = []
        for name, sub in node.items():
            parent_of[name] = parent
            if _is_branch(sub):
                children_of[name] = list(sub.keys())
                pending.append((sub, name))
            else:
                children_of.setdefault(name, [])
        stack.extend(reversed(pending))
    return levels, parent_of, children_of


def child_groups(levels, children_of):
    """Per level L>=1 the list of (parent_name, [child names]) for parents at
    L-1 that have children; channel order of level L is the concatenation."""
    out = []
    for L in range(1, len(levels)):
        groups = [(p, children_of.get(p, [])) for p in levels[L - 1]]
        out.append([(p, ch) for p, ch in groups if len(ch) > 0])
    return out


def get_classes(class_tree, full=False, final_counts=None):
    """Per-depth class counts: every node (``full``) or leaves only."""
    counts = []

    def walk(node, depth):
        if len(counts) <= depth:
            counts.append(0)
        for sub in node.values():
            branch = _is_branch(sub)
            if full or not branch:
                counts[depth] += 1
            if branch:
                walk(sub, depth + 1)

    walk(class_tree, 0)
    return counts


def level_order_names(tree):
    """BFS node names (parents before children) = target channel order
    (reference Data/dataset.py:70-86)."""
    order, queue = [], list(tree.items())
    while queue:
        name, sub = queue.pop(0)
        order.append(name)
        if _is_branch(sub):
            queue.extend(sub.items())
    return order


def leaf_names(tree):
    return [n for n in level_order_names(tree) if not _find(tree, n)]


def _find(tree, name):
    """children dict of ``name`` (empty dict for leaves)."""
    for k, v in tree.items():
        if k == name:
            return v if isinstance(v, dict) else {}
        if _is_branch(v):
            r = _find(v, name)
            if r is not None:
                return r
    return None



class TreeIndex:
    """One breadth-first pass over the class tree, shared by everything that walks it (predictEval's level synthesis,
    the target encoder): per node its depth, parent, direct children, leaf flag and the leaves below it.

    order     node names breadth first (= target channel order)
    levels    names per depth, in `order`
    children  name -> direct children (insertion order; [] for leaves)
    parent    name -> parent name (None at depth 0)
    depth     name -> depth
    leaves    name -> descendant leaves in depth-first order (a leaf maps to [itself])"""

    def __init__(self, tree):
        self.order, self.levels = [], []
        self.children, self.parent, self.depth = {}, {}, {}
        frontier = [(name, sub, None) for name, sub in tree.items()]
        d = 0
        while frontier:
            self.levels.append([name for name, _, _ in frontier])
            nxt = []
            for name, sub, par in frontier:
                self.order.append(name)
                self.parent[name], self.depth[name] = par, d
                kids = list(sub.keys()) if _is_branch(sub) else []
                self.children[name] = kids
                nxt.extend((k, sub[k], name) for k in kids)
            frontier, d = nxt, d + 1
        # leaves below every node: children are later in `order`, so one reverse sweep sees them first
        self.leaves = {}
        for name in reversed(self.order):
            kids = self.children[name]
            self.leaves[name] = [name] if not kids else [leaf for k in kids for leaf in self.leaves[k]]

    def is_leaf(self, name):
        return not self.children[name]

    @property
    def leaf_names(self):
        return [n for n in self.order if not self.children[n]]

    @property
    def parent_names(self):
        return [n for n in self.order if self.children[n]]
