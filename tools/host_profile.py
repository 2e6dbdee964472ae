"""cProfile of the host side of one eager train step (hier HRNet-W48, 620x620, B=4): where the Python time of the
~3,500 launches goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench

args = argparse.Namespace(gpus=1, steps=1, warmup=1, model="hrnet", batch=4, size=620, flat=False, tree="class_tree_tl.json")
device = torch.device("cuda", 0)
tree, model, ns, loss_fns, opt = bench.build(args, device)
from hrseg_amd import train as T
from hrseg_amd.utils import synth
x, t = synth.synthetic_batch(tree, 4, 620, seed=1, hierarchical=True)
x, t = torch.from_numpy(x).to(device), torch.from_numpy(t).to(device)
model.train()
ll = []
for _ in range(2):
    T.train_step(model, opt, x, t, loss_fns, ns, tree, ll)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
T.train_step(model, opt, x, t, loss_fns, ns, tree, ll)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
