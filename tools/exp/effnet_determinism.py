#!/usr/bin/env python3
"""Run-to-run determinism of the EfficientNet-B3 forward (fp16 storage): the same input through forward_blocks() several times, whole-block
kernels on and off; prints the number of differing elements per block boundary between repeats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.efficientnet import EfficientNet  # noqa: E402
from adafocus_amd.utils import nchw_to_nhwc4  # noqa: E402

dev = torch.device("cuda:0")
for size, image_size in ((100, "native"), (75, None), (144, "native")):
    m = EfficientNet.from_name("efficientnet-b3", num_classes=200, dtype="f16", image_size=image_size).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
    m = m.to(dev)
    g = torch.Generator().manual_seed(size)
    x4 = nchw_to_nhwc4((torch.randn((5, 3, size, size), generator=g) * 0.5).to(dev))
    for fusion in (True, False):
        m.fusion = fusion
        with torch.no_grad():
            for k in range(1, 27):
                ref = m.engine().forward_blocks(x4, k).clone()
                bad = 0
                for _ in range(6):
                    bad = max(bad, int((m.engine().forward_blocks(x4, k) != ref).sum()))
                if bad:
                    print("size %d pad %s fusion %d: %d elements differ between repeats after %d blocks" % (size, image_size, fusion, bad, k))
                    break
            else:
                print("size %d pad %s fusion %d: deterministic" % (size, image_size, fusion))
