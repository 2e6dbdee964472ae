"""Data-parallel train step end to end on the GPU: two ranks (two processes sharing the one card, gloo as the
transport -- RCCL refuses two ranks per device) run the engine's real reverse pass with GradSync's bucketed
all-reduce; the synchronised gradient must be the sum of the two ranks' local gradients, and after the AdamW
step (grad_scale = 1/world) both replicas must hold identical weights."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


SIZES = {"unet_hier_tl_62": 62, "hrnet_hier_tl_64": 128}      # HRNet at 128: its lowest branch still has 4x4 pixels
PER_RANK = 2


def _setup(name):
    import argparse
    from tests.helpers import CASES, build_model, level_weights_for, load_tree
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd.utils import synth
    kind, hier, tree_file, _, _ = CASES[name]
    size = SIZES[name]
    tree = load_tree(tree_file)
    num_classes = [4, 4]
    weights = level_weights_for(tree_file, hier)
    args = argparse.Namespace(model_type=1, model_select=0 if kind == "unet" else 1, num_classes=num_classes,
                              level_weights=weights, level0_pretrain_epochs=None, batch_size=PER_RANK)
    model = build_model(PM, kind, hier, tree, size).cuda()
    model.train()
    fns = [[PL.CrossEntropyLoss(), PL.SoftDiceLoss(num_classes=n)] for n in num_classes]
    x, t = synth.synthetic_batch(tree, 2 * PER_RANK, size, seed=17, hierarchical=True, blob=4)
    return model, args, tree, fns, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()


def _local_grad(name, shard):
    """flat gradient of one shard, no synchronisation"""
    from hrseg_amd import train as PT
    model, args, tree, fns, x, t = _setup(name)
    opt = PT.FusedAdamW(model, lr=[0.0])                  # lr 0: the step leaves the weights alone
    PT.train_step(model, opt, x[shard * PER_RANK:(shard + 1) * PER_RANK], t[shard * PER_RANK:(shard + 1) * PER_RANK], fns, args, tree, [])
    torch.cuda.synchronize()
    return model._flat.grad.clone()


def _worker(rank, world, port, name, out_dir, deterministic):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HRSEG_WGRAD_STREAM="0",           # two processes time-slice one card: keep one stream each
                      HRSEG_DETERMINISTIC="1" if deterministic else "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hrseg_amd import train as PT
    from hrseg_amd.parallel import GradSync
    model, args, tree, fns, x, t = _setup(name)
    with torch.no_grad():                                # replicas start different: construction must fix that
        for p in model.parameters():
            p.add_(0.01 * rank)
    sync = GradSync(model)
    opt = PT.FusedAdamW(model, lr=[1e-3])
    opt.grad_scale = 1.0 / world
    PT.train_step(model, opt, x[rank * PER_RANK:(rank + 1) * PER_RANK], t[rank * PER_RANK:(rank + 1) * PER_RANK], fns, args, tree, [])
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"grad{rank}.npy"), model._flat.grad.cpu().numpy())
    np.save(os.path.join(out_dir, f"data{rank}.npy"), model._flat.data.cpu().numpy())
    assert len(sync.launched) >= 2 and sync.launched[0][1] == model._flat.numel and sync.launched[-1][0] == 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("name", ["unet_hier_tl_62", "hrnet_hier_tl_64"])
def test_two_rank_step_sums_gradients_and_keeps_replicas_identical(name, deterministic, tmp_path):
    from hrseg_amd import _lib
    port = 29600 + (os.getpid() % 300) + (300 if deterministic else 0)
    mp.spawn(_worker, args=(2, port, name, str(tmp_path), deterministic), nprocs=2, join=True)
    _lib.set_deterministic(deterministic)
    try:
        _check(name, tmp_path, deterministic)
    finally:
        _lib.set_deterministic(False)


def _check(name, tmp_path, deterministic):
    g0, g1 = np.load(tmp_path / "grad0.npy"), np.load(tmp_path / "grad1.npy")
    d0, d1 = np.load(tmp_path / "data0.npy"), np.load(tmp_path / "data1.npy")
    assert np.array_equal(g0, g1), "all-reduced gradients differ between ranks"
    assert np.array_equal(d0, d1), "replicas diverged after the optimizer step"
    want = (_local_grad(name, 0) + _local_grad(name, 1)).cpu().numpy()
    again = (_local_grad(name, 0) + _local_grad(name, 1)).cpu().numpy()      # run-to-run noise of the fp32 atomics
    scale = np.abs(want).max()
    noise = np.abs(again - want).max() / scale
    if deterministic:
        # single-adder reductions everywhere: the local gradients reproduce bit for bit, and the all-reduced
        # gradient is their fp32 sum
        assert np.array_equal(again, want), "deterministic mode: two runs of the same shard differ"
        assert np.abs(g0 - want).max() / scale < 1e-5, np.abs(g0 - want).max() / scale
        return
    assert np.abs(g0 - want).max() / scale < max(2e-2, 4 * noise), (np.abs(g0 - want).max() / scale, noise)
    big = np.abs(want) > 1e-3 * scale
    med = np.median(np.abs(g0 - want)[big] / np.abs(want)[big])
    med_noise = np.median(np.abs(again - want)[big] / np.abs(want)[big])
    assert med < max(5e-3, 4 * med_noise), (med, med_noise)
