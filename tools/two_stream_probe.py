#!/usr/bin/env python3
"""Experiment: does running the trunk as two half-batches on two streams (co-scheduled kernels fill each
other's memory-bound phases and tile-quantisation tails) beat one full-batch pass?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth  # noqa: E402
from adafocus_amd.resnet import resnet50  # noqa: E402

dev = torch.device("cuda:0")
nets = []
for i in range(4):
    net = resnet50(num_classes=200).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
    nets.append(net.to(dev))
n, p = 1024, 96
x = torch.randn((n, p, p, 4), device=dev)
x[..., 3] = 0


def run(parts, stagger=False):
    chunks = x.chunk(parts)
    streams = [torch.cuda.Stream() for _ in range(parts)]
    outs = [None] * parts

    def once():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i] = nets[i].features_nhwc4(chunks[i])
        for s in streams:
            cur.wait_stream(s)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3


with torch.no_grad():
    for parts in (1, 2, 4):
        print("streams=%d  %.3f ms per 1024 patches" % (parts, run(parts)), flush=True)
