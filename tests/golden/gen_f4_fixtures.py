"""Oracle-side fixtures of the opt-in extensions, so that their HRNet GPU tests cost seconds instead of minutes
of CPU oracle time on the GPU box:

    python tests/golden/gen_f4_fixtures.py      # writes tests/golden/concat_oracle_hrnet_64.npz

The reference has no running code for these extensions (its level loop re-encodes the image only, models.py:267,277), so the
fixture comes from the repo's own CPU oracle twin (oracle/models.py, `concat_prev_logits=True`), evaluated in fp32, in fp64
and with level 1's input detached -- exactly what tests/helpers.concat_oracle_results computes live for the UNet case."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.helpers import concat_oracle_results  # noqa: E402

if __name__ == "__main__":
    import torch
    torch.set_num_threads(8)
    for kind, size in [("hrnet", 64)]:
        out = concat_oracle_results(kind, size)
        path = os.path.join(HERE, f"concat_oracle_{kind}_{size}.npz")
        np.savez_compressed(path, **out)
        print(path, f"{os.path.getsize(path) / 2**20:.2f} MiB")
