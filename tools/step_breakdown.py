#!/usr/bin/env python3
"""Where a bench step goes: HIP-event time of each stage of GFV.hot_path at the bench workload."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import act_args, synth_model_state  # noqa: E402
from adafocus_amd import hip_ops, synth  # noqa: E402
from adafocus_amd.gfv_net import GFV  # noqa: E402
from adafocus_amd.utils import get_patch_nhwc4  # noqa: E402

b, t, p = 64, 16, 96
dev = torch.device("cuda:0")
model = GFV(act_args(t, p, b)).eval()
model.load_state_dict(synth_model_state(model, 1007))
model = model.to(dev)
frames = torch.from_numpy(synth.synth_frames(b, t, 224, seed=100)).to(dev).view(b * t, 3, 224, 224)
_, act_np = synth.synth_actions(b * t, 7, seed=2)
actions = torch.from_numpy(act_np).to(dev)
gvec = torch.randn((b, t, 1280), device=dev)
feature = torch.empty((b, t, 3328), device=dev)
flat = feature.view(b * t, -1)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


patches = get_patch_nhwc4(frames, actions, p)
with torch.no_grad():
    print("gather          %.3f ms" % timeit(lambda: get_patch_nhwc4(frames, actions, p)))
    print("trunk           %.3f ms" % timeit(lambda: model.focuser.net.features_nhwc4(patches, out=flat[:, 1280:])))
    print("concat copy     %.3f ms" % timeit(lambda: hip_ops.copy2d(gvec.reshape(b * t, 1280), flat[:, :1280])))
    print("GRU + FC        %.3f ms" % timeit(lambda: model.classifier(feature)))
    print("hot_path total  %.3f ms" % timeit(lambda: model.hot_path(frames, gvec, actions, b, t)))
    scan = frames.view(b, t * 3, 224, 224)
    print("glancer         %.3f ms" % timeit(lambda: model.glancer.net.features_nhwc(frames), 3))
    fmap, fvec = model.glancer.net.features_nhwc(frames)
    table = model.focuser.action_table(dev)
    print("policy          %.3f ms" % timeit(lambda: model.focuser.policy.policy_old.act_sequence_nhwc(fmap, b, t, table), 3))
    print("full forward    %.3f ms" % timeit(lambda: model.offline_forward(scan, scan), 3))
