"""Host side of one train step (hier HRNet-W48, 620x620, B=4): cProfile of the eager step's Python, then the launch tape of the
same step entry by entry -- host time per C-ABI entry point (library planning + hipLaunchKernel calls), per stream wait and per
host callback, summed over one replay.  `python tools/host_profile.py [eager|tape|both]`"""
import cProfile, pstats, sys, os, io, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench

what = sys.argv[1] if len(sys.argv) > 1 else "both"
args = argparse.Namespace(gpus=1, steps=1, warmup=1, model=os.environ.get("MODEL", "hrnet"), batch=4, size=620, flat=False,
                          tree="class_tree_tl.json")
device = torch.device("cuda", 0)
tree, model, ns, loss_fns, opt = bench.build(args, device)
from hrseg_amd import _lib, train as T
from hrseg_amd.utils import synth
x, t = synth.synthetic_batch(tree, 4, 620, seed=1, hierarchical=True)
x, t = torch.from_numpy(x).to(device), torch.from_numpy(t).to(device)
model.train()
ll = []
for _ in range(2):
    T.train_step(model, opt, x, t, loss_fns, ns, tree, ll)
torch.cuda.synchronize()
if what in ("eager", "both"):
    pr = cProfile.Profile()
    pr.enable()
    T.train_step(model, opt, x, t, loss_fns, ns, tree, ll)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue())
if what in ("tape", "both"):
    step = T.TapedTrainStep(model, opt, loss_fns, ns, tree, x, t)
    for _ in range(2):
        step(x, t)
    torch.cuda.synchronize()
    tape = step.tape
    cur, rec = _lib.stream(), tape.main
    acc, cnt = collections.Counter(), collections.Counter()
    t_all = time.perf_counter()
    for kind, a, b, c in tape.entries:
        t0 = time.perf_counter()
        if kind == 0:
            a(*b, cur if c == rec else c)
            key = a.__name__
        elif kind == 1:
            c.record(b if b is not None else torch.cuda.current_stream())
            (a if a is not None else torch.cuda.current_stream()).wait_event(c)
            key = "(stream wait)"
        else:
            a()
            key = "(host callback)"
        acc[key] += time.perf_counter() - t0
        cnt[key] += 1
    total = time.perf_counter() - t_all
    torch.cuda.synchronize()
    print(f"one replay, timed entry by entry: {1e3 * total:.2f} ms for {len(tape.entries)} entries")
    for k, v in acc.most_common():
        print(f"  {1e3 * v:7.3f} ms  {cnt[k]:5d} x {1e6 * v / cnt[k]:7.1f} us  {k}")
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tape.replay()
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("plain replay(): " + ", ".join(f"{1e3 * v:.2f}" for v in ts) + " ms")
