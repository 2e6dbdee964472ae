"""Gradient-sync bucket logic with world_size 2 and 4 on CPU (gloo): every flat-gradient element is
all-reduced exactly once, in reverse registration order, as the marks of the last backward level pass."""
import os
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_model(rank):
    names = ["stem.0.weight", "layer1.0.conv1.weight", "transition1.0.0.weight", "stage2.0.w", "transition2.2.0.0.weight",
             "stage3.0.w", "transition3.3.0.0.weight", "stage4.0.w", "shared_head.0.weight", "classifiers.0.weight",
             "films.0.mlp.1.weight"]
    sizes = [100, 3000, 2000, 5000, 1500, 9000, 2500, 12000, 4000, 60, 40]
    slots, off = {}, 0
    for n, s in zip(names, sizes):
        slots[n] = (off, s)
        off += s
    flat = types.SimpleNamespace(slots=slots, numel=off, grad=torch.arange(off, dtype=torch.float32) * (rank + 1),
                                 data=torch.full((off,), float(rank + 7)))
    bufs = [torch.full((5,), float(rank)), torch.tensor(rank, dtype=torch.int64)]
    return types.SimpleNamespace(_flat=flat, _grad_hook=None, buffers=lambda: bufs, _bufs=bufs)


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hrseg_amd.parallel import GradSync
    model = _fake_model(rank)
    sync = GradSync(model, min_bucket=1000)
    assert model._grad_hook is sync
    # construction broadcast rank 0's parameters and buffers
    assert float(model._flat.data.min()) == float(model._flat.data.max()) == 7.0
    assert float(model._bufs[0].max()) == 0.0 and int(model._bufs[1]) == 0
    for step in range(2):
        model._flat.grad = torch.arange(model._flat.numel, dtype=torch.float32) * (rank + 1)
        # the reverse pass of the last level crosses the marks in this order
        for mark in ("shared_head", "transition3", "transition2", "transition1", "layer1", "end"):
            sync(mark)
        expect = torch.arange(model._flat.numel, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert torch.equal(model._flat.grad, expect), f"rank {rank} step {step}"
        # contiguous, non-overlapping, descending, complete
        spans = sync.launched
        assert spans[0][1] == model._flat.numel and spans[-1][0] == 0
        for (lo, hi), (lo2, hi2) in zip(spans, spans[1:]):
            assert hi2 == lo and lo2 < hi2
        assert len(spans) >= 4
    # global metrics: per-level confusion counts summed over the ranks = counts of the gathered batch
    from hrseg_amd.parallel import all_reduce_confusion
    from hrseg_amd.Metrics.performance_metrics import metrics_from_confusion
    g = torch.Generator().manual_seed(5)
    parts = [[torch.randint(0, 50, (4, 4), generator=g), torch.randint(0, 50, (5, 5), generator=g)] for _ in range(world)]
    got = all_reduce_confusion([c.clone() for c in parts[rank]])
    for L, c in enumerate(got):
        want = sum(parts[r][L] for r in range(world))
        assert c.dtype == torch.int64 and torch.equal(c, want)
        m = metrics_from_confusion(c, child_classes=(L > 0))
        assert torch.allclose(m["iou"], metrics_from_confusion(want, child_classes=(L > 0))["iou"])
    # Dice under data parallelism: the divisor is the GLOBAL number of valid items (reference losses.py:64-66 on the
    # gathered batch).  Rank 0: two items, both valid; rank 1: two items, one all-ignored at this level.
    from hrseg_amd.Metrics.losses import global_batch_dice
    # World 4 adds a rank without any valid item (its Dice term and gradient vanish, the others carry the whole mean).
    items = {2: [[0.30, 0.50], [0.80]], 4: [[0.30, 0.50], [0.80], [], [0.10, 0.20, 0.60]]}[world]
    item_dice = items[rank]                                       # per-item Dice values of the rank's valid items
    local = torch.tensor(item_dice, dtype=torch.float64, requires_grad=True)
    res = [torch.zeros(()), (local.sum() / max(len(item_dice), 1)).float(), torch.tensor(float(len(item_dice)))]
    term = global_batch_dice(res)
    flat_items = [v for r in items for v in r]
    gathered = sum(flat_items) / len(flat_items)
    avg = term.detach().clone().double().reshape(1)
    dist.all_reduce(avg)
    assert abs(float(avg) / world - gathered) < 1e-6, (float(avg) / world, gathered)       # mean over ranks = gathered Dice
    term.backward()
    # gradient averaged over the ranks: d(gathered)/d(item) = 1/n_global for every valid item
    assert torch.allclose(local.grad / world, torch.full_like(local, 1.0 / len(flat_items)), atol=1e-6), local.grad
    # no valid item anywhere: the term is zero, not NaN
    z = global_batch_dice([torch.zeros(()), torch.zeros((), requires_grad=True) * 1.0, torch.tensor(0.0)])
    assert float(z) == 0.0
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)


def test_bucketed_allreduce_world4_gloo():
    """four ranks: bucket boundaries / completeness of the exchange, summed confusion counts, and the global Dice divisor
    with one rank that holds no valid item at all"""
    port = 30600 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(4, port), nprocs=4, join=True)


def test_single_process_is_a_noop():
    from hrseg_amd.parallel import GradSync
    model = _fake_model(0)
    before = model._flat.grad.clone()
    sync = GradSync(model)
    for mark in ("shared_head", "end"):
        sync(mark)
    assert torch.equal(model._flat.grad, before) and sync.launched[-1][0] == 0
