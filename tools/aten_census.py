"""Which launches of one train step are NOT this package's kernels (ATen elementwise / fill / copy kernels, runtime buffer
copies), and which line of the package issues them: torch.profiler with Python stacks over one eager step of the headline
configuration (hier HRNet-W48, 620x620, B=4).  Every such launch costs its own ~2 us plus a kernel boundary."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
from torch.profiler import profile, ProfilerActivity
import bench

args = argparse.Namespace(gpus=1, steps=1, warmup=1, model=os.environ.get("MODEL", "hrnet"), batch=4, size=620, flat=False,
                          tree="class_tree_tl.json")
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    loss, cms = T.train_step(model, opt, x, t, loss_fns, ns, tree, ll)
    vec = T._metric_vectors(cms)                       # the rest of bench.py's batch body
    host = torch.cat([loss.reshape(1)] + [vec[k] for k in T.METRIC_NAMES]).tolist()
    torch.cuda.synchronize()
PKG = os.sep + "restrictive-hierarchical-semantic-segmentation_amd" + os.sep
sites = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    ndev = sum(1 for k in ev.kernels) if hasattr(ev, "kernels") else 0
    if not ndev:
        continue
    frame = next((s for s in (ev.stack or []) if PKG in s or "hrseg_amd" in s or "aten_census" in s), "(outside the package)")
    sites[(frame.strip()[-110:], ev.name)] += ndev
copies = collections.Counter()
for ev in prof.events():
    if "emcpy" in ev.name or "emset" in ev.name:
        par, frame = ev.cpu_parent, None
        while par is not None and frame is None:
            frame = next((s for s in (par.stack or []) if PKG in s or "hrseg_amd" in s or "bench.py" in s), None)
            par = par.cpu_parent
        copies[((frame or "(no package frame)").strip()[-110:], ev.name)] += 1
print(f"{sum(copies.values())} runtime memcpy / memset calls")
for (frame, name), n in copies.most_common(40):
    print(f"{n:5d}  {name:28s} {frame}")
total = sum(sites.values())
print(f"{total} device launches from ATen ops in one step")
for (frame, name), n in sites.most_common(60):
    print(f"{n:5d}  {name:28s} {frame}")
