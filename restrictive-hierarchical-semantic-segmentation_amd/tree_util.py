"""Tree helpers with the reference's tree_util.py call surface (tree_util.py:6-140).

Pure host-side Python; the training loop imports these names but never calls
them, so they are provided for drop-in imports.
"""
from __future__ import annotations


class node:
    def __init__(self, name):
        self.name = name
        self.children = []
        self.channel = None
        self.level = None


def create_tree_from_textfile(filename):
    """Tab-indented text file -> tree under a synthetic root; indentation may grow by one per line."""
    root = node("Universal class")
    stack, depth, prev = [root], 0, None
    with open(filename, "r") as fd:
        for line in fd:
            tabs = line.count("\t")
            new = node(line.strip())
            if tabs == depth + 1:
                stack.append(prev)
                depth += 1
            elif tabs < depth:
                while depth > tabs:
                    stack.pop()
                    depth -= 1
            elif tabs != depth:
                raise RuntimeError("Indentation can only increase by one")
            stack[-1].children.append(new)
            prev = new
    return root


def add_channels(node, channel):
    """number the leaves depth-first; returns the next free channel"""
    if not node.children:
        node.channel = channel
        return channel + 1
    for child in node.children:
        channel = add_channels(child, channel)
    return channel


def update_channels(node, class_lookup):
    if not node.children:
        node.channel = class_lookup[node.channel]
        return
    for child in node.children:
        update_channels(child, class_lookup)


def add_levels(node, depth):
    if not node.children:
        node.level = depth - 1
        return
    for child in node.children:
        child.level = depth - 1
        if child.children:
            add_levels(child, depth - 1)


def getLeafClasses(node, my_list):
    if not node.children:
        my_list.append(node.channel)
        return my_list
    for child in node.children:
        getLeafClasses(child, my_list)
    return my_list


def find_depth(node):
    if not node.children:
        return 0
    return 1 + max(find_depth(c) for c in node.children)


def getLossLevelList(root, level, myList):
    for child in root.children:
        if not child.children or child.level == level:
            myList.append(getLeafClasses(child, []))
        else:
            getLossLevelList(child, level, myList)


def getTreeList(node):
    out = []
    for level in range(find_depth(node)):
        level_list = []
        getLossLevelList(node, level, level_list)
        out.append(level_list)
    return out
