"""one shape through the wide-tile im2col body, a few launches (for rocprofv3 --pmc): python tools/spw_one.py [wide]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrseg_amd import _lib, ops  # noqa: E402

wide = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
x = torch.randn(8, 155, 155, 720, device=dev)
w = torch.randn(720, 1, 720, device=dev) * 0.05
_lib.tune(sp_wide=wide)
for _ in range(5):
    ops.conv_fwd(x, w, None, 1, 1, prec=_lib.CONV_PRECISION["auto"])
torch.cuda.synchronize()
