#!/usr/bin/env python3
"""One sha256 per case of EfficientNet-B3's first blocks (whose narrow gated project convs -- K <= 64, N <= 32 -- take the
persistent strip kernel ef_nproj_kernel by default).  tests/test_effnet.py calls digests() with the library option "effnet_plan"
at its default and with ADAF_EF_PLAN_STRIP_PROJECT cleared (gated_project_kernel) and expects the same digests."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import _lib, synth  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402


def digests():
    dev = torch.device("cuda:0")
    rows = []
    for dt in ("f32", "f16"):
        m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype=dt).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
        m = m.to(dev)
        net = m.engine()
        for n, p in ((260, 144), (300, 100)):          # 260 x 72^2 rows: enough strips for the persistent grid; odd maps at 100^2
            g = torch.Generator().manual_seed(n + p)
            x = (torch.randn((n, p, p, 4), generator=g) * 0.8).to(dev)
            x[..., 3] = 0
            with torch.no_grad():
                for upto in (1, 2):
                    y = net.forward_blocks(x, upto)
                    rows.append((dt, n, p, upto, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()))
    return rows


if __name__ == "__main__":
    plan = int(_lib.get_option("effnet_plan"))
    for label, value in (("strip kernel", plan | _lib.EF_PLAN_STRIP_PROJECT), ("tiled kernel", plan & ~_lib.EF_PLAN_STRIP_PROJECT)):
        with _lib.option("effnet_plan", value):
            for row in digests():
                print(label, *row)
