"""Does a train step retain device memory?  memory_allocated() after each step + the largest live tensors."""
import sys, os, gc, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hrseg_amd import train as T
from hrseg_amd.utils import synth

args = argparse.Namespace(gpus=1, steps=1, warmup=1, model="hrnet", batch=4, size=320, flat=False, tree="class_tree_tl.json")
device = torch.device("cuda", 0)
tree, model, ns, loss_fns, opt = bench.build(args, device)
x, t = synth.synthetic_batch(tree, 4, 320, seed=1, hierarchical=True)
x, t = torch.from_numpy(x).to(device), torch.from_numpy(t).to(device)
model.train()
ll = []
for i in range(8):
    loss, cms = T.train_step(model, opt, x, t, loss_fns, ns, tree, ll)
    torch.cuda.synchronize()
    gc.collect()
    print("step %d: allocated %.1f MB, reserved %.1f MB, peak %.1f MB" % (
        i, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, torch.cuda.max_memory_allocated() / 2**20))
live = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda]
from collections import Counter
c = Counter((tuple(o.shape), str(o.dtype)) for o in live)
big = sorted(c.items(), key=lambda kv: -kv[1] * int(torch.tensor(kv[0][0]).prod() if kv[0][0] else 1))[:12]
for (shape, dt), n in big:
    print(n, shape, dt)
