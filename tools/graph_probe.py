"""Does a captured hipMemsetAsync / torch.zeros behave under hipGraph replay on this box?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hrseg_amd import ops
buf = torch.full((1 << 20,), 7.0, device="cuda")
z = torch.randn(2, 4, 64, 64, device="cuda")
t = (torch.rand(2, 4, 64, 64, device="cuda") > 0.5).float()
w = torch.ones(4, device="cuda")
ref, _ = ops.loss_fwd(z, t, w)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    buf.zero_()
    buf.add_(1)
    out, coef = ops.loss_fwd(z, t, w)
    zz = torch.zeros(1000, device="cuda")
    zz.add_(2)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "buf", buf[:3].tolist(), float(buf.sum()) / buf.numel(), "loss", out.tolist(), "ref", ref.tolist(),
          "zz", float(zz.sum()) / 1000)
