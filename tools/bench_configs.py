#!/usr/bin/env python3
"""Hot-path throughput of the other BASELINE.json configurations (parity-test cases, not bench lines):
C3 = ActivityNet T=16, P=128;  C4 = Something-Something TSM-ResNet-50, T=8, P=128 (per GPU, B=64)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import act_args, synth_model_state  # noqa: E402
from adafocus_amd import synth  # noqa: E402

dev = torch.device("cuda:0")


def clock(fn, steps=10, warm=3):
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for i in range(warm):
        with torch.cuda.stream(streams[i % 2]):
            fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % 2]):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def act(t, p, b=64):
    from adafocus_amd.gfv_net import GFV
    m = GFV(act_args(t, p, b)).eval()
    m.load_state_dict(synth_model_state(m, 1007))
    m = m.to(dev)
    frames = torch.from_numpy(synth.synth_frames(b, t, 224, seed=1)).to(dev).view(b * t, 3, 224, 224)
    actions = torch.from_numpy(synth.synth_actions(b * t, 7, seed=2)[1]).to(dev)
    gvec = torch.randn((b, t, 1280), device=dev)
    with torch.no_grad():
        sec = clock(lambda: m.hot_path(frames, gvec, actions, b, t))
    flop = 2 * {96: 0.7507e9, 128: 1.3346e9}[p] * b * t
    print("ACT  T=%d P=%d B=%d: %.2f ms/step  %.0f clips/s  (local CNN %.1f TFLOP/s incl. gather+GRU time)" %
          (t, p, b, sec * 1e3, b / sec, flop / sec / 1e12), flush=True)


def sth(t=8, p=128, b=64):
    from adafocus_amd.gfv_net_sth import GFV
    from tests.test_state_dict_compat import sth_args
    a = sth_args()
    a.gpu, a.batch_size = 0, b
    m = GFV(a).eval()
    m.load_state_dict(synth_model_state(m, 1007))
    m = m.to(dev)
    fo = torch.from_numpy(synth.synth_frames(b, t, 224, seed=4)).view(b, t, 3, 224, 224).to(dev)
    fm = torch.randn((b, t, 7, 7, 1280), device=dev).permute(0, 1, 4, 2, 3)     # glancer map, reference-layout view
    glog = torch.randn((b, t, 174), device=dev)
    forced = torch.rand((b, 2), device=dev)
    with torch.no_grad():
        sec = clock(lambda: m.action_stage3(fo, fm, glog, 0, a, prev_local_patch=None, forced_action=forced))
    print("STH  TSM-R50 T=%d P=%d B=%d (gather + TSM trunk + FC/mean, action given): %.2f ms/step  %.0f clips/s  (%.1f TFLOP/s)" %
          (t, p, b, sec * 1e3, b / sec, 2 * 1.3346e9 * b * t / sec / 1e12), flush=True)


act(16, 96)
act(8, 96)
act(16, 128)
sth()
