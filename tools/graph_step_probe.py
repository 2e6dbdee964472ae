"""Eager vs hipGraph-replayed train steps on small models: losses must track each other."""
import os, sys, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import train as T
from hrseg_amd.Metrics import losses
from hrseg_amd.Models import models
from hrseg_amd.utils import synth
from hrseg_amd.utils.config import hrnet_w48_config
from hrseg_amd.utils.hierarchy import get_classes
from tests.helpers import load_tree

kind = sys.argv[1] if len(sys.argv) > 1 else "unet"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tree = load_tree("class_tree_tl.json")
nc = get_classes(tree, full=True)
w = synth.README_LEVEL_WEIGHTS_TL
x, t = synth.synthetic_batch(tree, 2, size, seed=3, hierarchical=True, blob=4)
x, t = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
args = argparse.Namespace(model_type=1, model_select=0 if kind == "unet" else 1, num_classes=nc, level_weights=w,
                          level0_pretrain_epochs=None, batch_size=2)
def make():
    m = models.UNet(size=size, n_channels=3, hierarchy=tree, model_type=1) if kind == "unet" else \
        models.HighResolutionNet(hrnet_w48_config(), hierarchy=tree, model_type=1)
    m = synth.fill_state_dict(m).cuda()
    m.train()
    return m, T.FusedAdamW(m, lr=[1e-4]), [[losses.CrossEntropyLoss(), losses.SoftDiceLoss(num_classes=n)] for n in nc]
m, opt, fns = make()
eager = [float(T.train_step(m, opt, x, t, fns, args, tree, [])[0]) for _ in range(5)]
print("eager  ", ["%.5f" % v for v in eager])
m, opt, fns = make()
g = T.GraphedTrainStep(m, opt, fns, args, tree, x, t, warmup=1)     # steps 0 (eager) and 1 (capture: not executed!)
out = []
for _ in range(4):
    loss, _ = g(x, t)
    out.append(float(loss))
    gn = float(m._flat.grad.norm())
print("graphed", ["%.5f" % v for v in out], "grad norm", gn, "adam state", opt._state.tolist())
