#!/usr/bin/env python3
"""Small-batch latency of the hot path (gather -> ResNet-50 -> GRU classifier): launches issued one by one from Python
(hot_path) vs the same step replayed from a captured HIP graph (GFV.capture_hot_path).  B = 1, 2 at T = 8 (BASELINE config
1 is B = 2, T = 8, P = 96) and T = 16.  The rows bench.py reports as `also.latency_small_batch`."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_extras as X  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(X.latency_rows(torch.device("cuda:0"))))
