"""Which part of the step misbehaves under hipGraph replay?  forward-only, then forward+backward."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import train as T
from hrseg_amd.Metrics import losses
from hrseg_amd.Models import models
from hrseg_amd.utils import synth
from hrseg_amd.utils.hierarchy import get_classes
from tests.helpers import load_tree
size = 32
tree = load_tree("class_tree_tl.json")
nc = get_classes(tree, full=True)
w = synth.README_LEVEL_WEIGHTS_TL
x, t = synth.synthetic_batch(tree, 2, size, seed=3, hierarchical=True, blob=4)
x, t = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
m = synth.fill_state_dict(models.UNet(size=size, n_channels=3, hierarchy=tree, model_type=1)).cuda()
m.train()
ts = [t[:, :4].contiguous(), t[:, 4:].contiguous()]

def fwd_bwd():
    m.zero_grad()
    probs, logits = m(x, type=1)
    loss = 0.0
    for L in range(2):
        r = losses.fused_ce_dice(logits[L], ts[L], w[L])
        loss = loss + r[0] + r[1]
    loss.backward()
    return loss.detach(), [z.detach() for z in logits]

# eager reference (BN running stats change, outputs in train mode do not depend on them)
l0, z0 = fwd_bwd()
g0 = m._flat.grad.clone()
l1, z1 = fwd_bwd()
print("eager repeat: loss", float(l0), float(l1), "grad diff", float((m._flat.grad - g0).abs().max()))

# A: forward only under graph
with torch.no_grad():
    m(x, type=1)
torch.cuda.synchronize()
gA = torch.cuda.CUDAGraph()
with torch.cuda.graph(gA):
    with torch.no_grad():
        pA, zA = m(x, type=1)
for i in range(2):
    gA.replay(); torch.cuda.synchronize()
    print("A fwd replay", i, "logit diff", [float((a - b).abs().max()) for a, b in zip(zA, z0)])

# B: forward + backward under graph
gB = torch.cuda.CUDAGraph()
with torch.cuda.graph(gB):
    lB, zB = fwd_bwd()
for i in range(3):
    gB.replay(); torch.cuda.synchronize()
    print("B replay", i, "loss", float(lB), "logit diff", [float((a - b).abs().max()) for a, b in zip(zB, z0)],
          "grad diff", float((m._flat.grad - g0).abs().max()), "grad norm", float(m._flat.grad.norm()), float(g0.norm()))

# C: what persistent state does a fwd+bwd replay modify?
print("--- C")
import copy
snap = lambda: {"data": m._flat.data.clone(), **{n: b.clone() for n, b in m.named_buffers()}}
s0 = snap()
gB.replay(); torch.cuda.synchronize()
s1 = snap()
changed = [(k, float((s0[k].double() - s1[k].double()).abs().max())) for k in s0 if not torch.equal(s0[k], s1[k])]
print("changed by one replay:", [(k, v) for k, v in changed if "running" not in k and "num_batches" not in k][:10],
      "| n running-stat buffers changed:", sum(1 for k, _ in changed if "running" in k))
print("x changed?", float((x - torch.from_numpy(synth.synthetic_batch(tree, 2, size, seed=3, hierarchical=True, blob=4)[0]).cuda()).abs().max()))

# D: same fwd+bwd but WITHOUT torch.autograd (engine reverse pass called from this thread)
print("--- D")
from hrseg_amd import ops
ones = torch.ones(3, device="cuda")
wts = [torch.tensor(w[L], device="cuda") for L in range(2)]
def manual():
    ops.fill(m._flat.grad, 0.0)
    m._flat.grads_fresh = True
    run = m._run(x, True)
    dl, loss = [], 0.0
    for L in range(2):
        out, coef = ops.loss_fwd(run.logits[L], ts[L], wts[L])
        dl.append(ops.loss_bwd(run.logits[L], ts[L], coef, ones))
        loss = loss + out[0] + out[1]
    run.backward([None, None], dl)
    return loss, run.logits
lD, zD = manual()
torch.cuda.synchronize()
print("eager manual: loss", float(lD), "grad diff", float((m._flat.grad - g0).abs().max()))
gD = torch.cuda.CUDAGraph()
with torch.cuda.graph(gD):
    lD, zD = manual()
for i in range(4):
    gD.replay(); torch.cuda.synchronize()
    print("D replay", i, "loss", float(lD), "logit diff", [float((a - b).abs().max()) for a, b in zip(zD, z0)],
          "grad diff", float((m._flat.grad - g0).abs().max()))
