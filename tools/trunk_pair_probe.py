#!/usr/bin/env python3
"""ResNet-50 trunk, 1024 patches of P^2: one pass on one stream vs k sub-batches on k streams (does what helped the two MBConv networks --
adaf_effnet / adaf_mobilenetv2 pair their chunks -- help the trunk, whose launches are long?).  usage: trunk_pair_probe.py [P=96] [n=1024]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
p = int(sys.argv[1]) if len(sys.argv) > 1 else 96
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
x = torch.randn((n, p, p, 4), device=dev)
x[..., 3] = 0
ref = None
with torch.no_grad():
    for rnd in range(2):
        for k in (1, 2, 3, 4):
            streams = [torch.cuda.Stream() for _ in range(k)]
            cuts = [(i * n) // k for i in range(k + 1)]
            parts = [x[cuts[i]:cuts[i + 1]] for i in range(k)]
            outs = [None] * k

            def run():
                cur = torch.cuda.current_stream()
                for i, s in enumerate(streams):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        outs[i] = net.features_nhwc4(parts[i])
                for s in streams:
                    cur.wait_stream(s)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            full = torch.cat(outs)
            if ref is None:
                ref = full
            print("round %d: %d stream(s): %.3f ms per %d patches of %d^2; equal to the single pass: %s" % (rnd, k, e0.elapsed_time(e1) / 10, n, p, torch.equal(full, ref)), flush=True)
