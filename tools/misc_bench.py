"""Times the non-GEMM kernels of the step in isolation at the bench shapes (hier HRNet-W48, 620x620, B=4)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hrseg_amd
from hrseg_amd import ops, _lib


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


dev = "cuda"
B, H = 4, 620
z = torch.randn(B, 3, H, H, device=dev)
dp = torch.randn(B, 3, H, H, device=dev)
dz = torch.empty_like(z)
print("sigmoid_bwd contiguous dp        %8.1f us" % timeit(lambda: ops.sigmoid_bwd(dp, z, dz=dz, accumulate=False)))
print("sigmoid_bwd accumulate           %8.1f us" % timeit(lambda: ops.sigmoid_bwd(dp, z, dz=dz, accumulate=True)))
dpb = torch.randn(B, 3, device=dev)[:, :, None, None].expand(B, 3, H, H)
print("sigmoid_bwd broadcast dp         %8.1f us" % timeit(lambda: ops.sigmoid_bwd(dpb, z, dz=dz, accumulate=True)))

# stem weight gradient: x [4,620,620,3] -> dy [4,310,310,64], 3x3 stride 2
x = torch.randn(B, H, H, 3, device=dev)
dy = torch.randn(B, 310, 310, 64, device=dev)
dw = torch.zeros(64, 3, 3, 3, device=dev)
print("stem wgrad (3->64, s2)           %8.1f us" % timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 2)))
xu = torch.randn(B, H, H, 3, device=dev)
dyu = torch.randn(B, H, H, 64, device=dev)
print("unet inc wgrad (3->64, s1)       %8.1f us" % timeit(lambda: ops.conv_wgrad(xu, dyu, dw, 3, 1)))

for nb in (64, 128, 256, 512, 1024, 2048):
    _lib.set_wgrad_tune(0, 0, nb)
    print("  target blocks %5d: stem %8.1f us   unet inc %8.1f us" % (nb, timeit(lambda: ops.conv_wgrad(x, dy, dw, 3, 2)), timeit(lambda: ops.conv_wgrad(xu, dyu, dw, 3, 1))))
_lib.set_wgrad_tune(0, 0, 0)
