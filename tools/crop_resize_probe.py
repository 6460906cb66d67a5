#!/usr/bin/env python3
"""The resampling gather (adaf_crop_resize_f32, row N1) on 1024 frames of 224^2: HIP-event time and algorithmic GB/s at
S = 128, 192 and mixed window sizes -> 96^2 patches (the `gather_resize` rows of bench.py); run under rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE for the traffic behind them (tools/profile_r3.sh)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_extras as X  # noqa: E402

dev = torch.device("cuda:0")
frames = torch.randn((1024, 3, 224, 224), device=dev)
print(json.dumps(X.gather_resize_row(dev, frames, 96, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20)))
