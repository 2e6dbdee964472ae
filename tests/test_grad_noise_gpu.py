"""The gradient bars of test_models_gpu.py rest on one claim: on these small nets the fp32 CPU reference is itself ~1e-2 from
an exact evaluation in the earliest layers, and the GPU path is no further away.  This file checks that claim instead of
assuming it: every parameter gradient of the GPU path and of the fp32 CPU oracle is compared with an fp64 evaluation of the
oracle on the same golden inputs (the former diagnostic tests/diagnostics/grad_noise.py, now asserted)."""
import argparse

import numpy as np
import pytest
import torch

from tests.helpers import CASES, build_model, conv_mode, level_weights_for, load_golden, load_tree

pytestmark = pytest.mark.gpu
_FP64 = {}


def _oracle_grads(name, dtype):
    from oracle import models as OM, train_step as OT
    kind, hier, tree_file, size, batch = CASES[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    nc = [int(v) for v in g["num_classes"]]
    w = level_weights_for(tree_file, hier)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    m = build_model(OM, kind, hier, tree, size).to(dtype)
    m.train()
    out = OT.forward_loss(m, x.to(dtype), t.to(dtype), nc, w, hierarchical=hier, is_unet=(kind == "unet"), with_metrics=False)
    out["loss"].backward()
    return {n: p.grad.double().numpy() for n, p in m.named_parameters()}


@pytest.mark.parametrize("mode", ["auto", "auto_ws"])
@pytest.mark.parametrize("name", ["hrnet_flat_tl_64", "unet_hier_tl_62"])
def test_gpu_gradients_are_as_close_to_fp64_as_the_fp32_reference(name, mode):
    from hrseg_amd.Models import models as PM
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    kind, hier, tree_file, size, batch = CASES[name]
    if name not in _FP64:
        _FP64[name] = (_oracle_grads(name, torch.float64), _oracle_grads(name, torch.float32))
    g64, g32 = _FP64[name]
    g = load_golden(name)
    tree = load_tree(tree_file)
    nc = [int(v) for v in g["num_classes"]]
    w = level_weights_for(tree_file, hier)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["target"])
    args = argparse.Namespace(model_type=1 if hier else 0, model_select=0 if kind == "unet" else 1, num_classes=nc,
                              level_weights=w, level0_pretrain_epochs=None, batch_size=batch)
    pm = build_model(PM, kind, hier, tree, size).cuda()
    pm.train()
    # Deterministic mode: the claim is about OPERAND precision.  With atomics on, the summation order of the 64-pixel HRNet's
    # 2 x 2-pixel branch (BatchNorm over 8 samples) moves the result between two basins from run to run -- median 5.5e-3 or
    # 2.0e-2, worst parameter 0.15 or 0.24, whatever the kernels (measured round 4, three runs each at two nine-tap block
    # plans) -- which says nothing about 22-bit operands; the deterministic evaluation is the same every run (1.45e-3 / 9.6e-3).
    from hrseg_amd import _lib
    _lib.set_deterministic(True)
    try:
        e_gpu, e_cpu = _errors(pm, mode, x, t, args, tree, hier, w, g64, g32)
    finally:
        _lib.set_deterministic(False)
    _check(name, mode, e_gpu, e_cpu)


def _errors(pm, mode, x, t, args, tree, hier, w, g64, g32):
    from hrseg_amd.Metrics import losses as PL
    from hrseg_amd import train as PT
    with conv_mode(pm, mode):
        probs, logits = PT._model_call(pm, x.cuda(), args, tree)
        logits = logits if hier else [logits]
        targets = PT.split_targets(t.cuda(), args)
        loss = 0.0
        for L, (z, tt) in enumerate(zip(logits, targets)):
            r = PL.fused_ce_dice(z, tt, w[L])
            loss = loss + r[0] + r[1]
        loss.backward()
    e_gpu, e_cpu = [], []
    for n, p in pm.named_parameters():
        ref = g64[n]
        s = np.abs(ref).max()
        if s <= 1e-4:                       # gradients that are themselves rounding noise say nothing
            continue
        e_cpu.append(np.abs(g32[n] - ref).max() / s)
        e_gpu.append(np.abs(p.grad.cpu().double().numpy() - ref).max() / s)
    return np.array(e_gpu), np.array(e_cpu)


def _check(name, mode, e_gpu, e_cpu):
    assert len(e_gpu) > 20
    print(f"{name} {mode}: median err gpu {np.median(e_gpu):.2e} cpu32 {np.median(e_cpu):.2e}; max gpu {e_gpu.max():.2e} cpu32 {e_cpu.max():.2e}")
    # Default routing (`auto`: at these sizes the exact-fp32 MFMA kernels, fp16x2 weight gradients): the GPU path is as far from
    # fp64 as the fp32 CPU reference is -- measured medians 4.6-4.8e-3 vs 4.7e-3 (HRNet), 2.4e-3 vs 2.9e-3 (UNet); the worst
    # parameter 3.7e-2 ... 6.9e-2 vs 3.7e-2 from run to run (atomic summation order; it sits behind the 2 x 2-pixel branch whose
    # BatchNorm normalises over 8 samples, where every evaluation is ill-conditioned).
    # `auto_ws` forces EVERY contraction of these 64-pixel nets onto the fp16x2 kernels (22-bit operands instead of 24): the
    # HRNet gradients are then 2-3.5x (median 0.96-1.7e-2) and on the worst parameter 4-8x (0.15-0.29) further from fp64 than
    # the reference; UNet shows no difference (2.3e-3).  That is the price of 22-bit operands on ill-conditioned small layers and
    # the reason the default policy keeps problems under 8192 pixels on the exact-fp32 kernels; at the headline size the
    # fp16x2 kernels run only on the large layers (test_train_steps_track_the_oracle_at_256 pins that routing).
    # (deterministic evaluation, round 4: auto 0.31x / 0.60x of the reference at the median / maximum; auto_ws 2.0x / 4.2x)
    f_med, f_p90, f_max = (1.5, 2.0, 3.0) if mode == "auto" else (3.0, 6.0, 6.0)
    assert np.median(e_gpu) <= f_med * np.median(e_cpu) + 1e-6
    assert np.percentile(e_gpu, 90) <= f_p90 * np.percentile(e_cpu, 90) + 1e-6
    assert e_gpu.max() <= f_max * e_cpu.max() + 1e-5
