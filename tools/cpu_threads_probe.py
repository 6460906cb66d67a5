import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from bench import *
from adafocus_amd.gfv_net import GFV
from adafocus_amd import synth
from oracle import ref_model as O
m = GFV(act_args(16,96,4)).eval(); sd = synth_model_state(m, 1007)
frames = torch.from_numpy(synth.synth_frames(4, 16, 224, seed=1)).view(64, 3, 224, 224)
_, actions = synth.synth_actions(64, 7, seed=2); gvec = torch.randn(4, 16, 1280)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    ts = []
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter(); O.act_hot_path(sd, frames, gvec, torch.from_numpy(actions), 96); ts.append(time.perf_counter() - t0)
    print(th, "threads: %.2f s  %.2f clips/s" % (min(ts[1:]), 4 / min(ts[1:])), flush=True)
