"""CPU pins of the opt-in extensions' oracle twins.  The reference ships `grouped_conditional_kl` only as
commented-out text (Metrics/losses.py:180-210), so there is nothing to import or run: the oracle restatement is pinned by
hand-computed known answers."""
import math

import torch

from oracle import losses as OL


def _kl(z, p_parent=0.5, groups=(("p", ["a", "b"]),), levels_prev=("p",)):
    z = torch.tensor(z, dtype=torch.float32).reshape(1, -1, 1, 1)
    pp = torch.full((1, len(levels_prev), 1, 1), p_parent)
    return float(OL.grouped_conditional_kl(z, pp, list(groups), list(levels_prev)))


def test_grouped_kl_known_answers():
    # uniform children: KL(Q || U) = 0
    assert abs(_kl([0.3, 0.3])) < 1e-7
    # Q = (0.75, 0.25): 0.75 ln 1.5 + 0.25 ln 0.5, `.mean()` over the two channels
    want = (0.75 * math.log(1.5) + 0.25 * math.log(0.5)) / 2
    assert abs(_kl([math.log(3.0), 0.0]) - want) < 1e-6
    # the parent's probability is a log-bias common to the group: no effect
    assert abs(_kl([math.log(3.0), 0.0], p_parent=0.01) - want) < 1e-6
    # a saturated child: Q -> one-hot, KL -> ln g (mean over g channels), the clamp keeps log finite
    assert abs(_kl([60.0, 0.0, 0.0, 0.0], groups=(("p", list("abcd")),)) - math.log(4.0) / 4) < 1e-5
    # two groups (sizes 2 and 3) of different parents: the mean over groups of the per-group means
    a = (0.75 * math.log(1.5) + 0.25 * math.log(0.5)) / 2
    got = _kl([math.log(3.0), 0.0, 0.1, 0.1, 0.1], groups=(("p", ["a", "b"]), ("q", ["c", "d", "e"])), levels_prev=("p", "q"))
    assert abs(got - a / 2) < 1e-6
    # a parent without children is skipped
    got = _kl([math.log(3.0), 0.0], groups=(("p", ["a", "b"]), ("q", [])), levels_prev=("p", "q"))
    assert abs(got - a) < 1e-6
