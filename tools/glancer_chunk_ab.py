import os, sys, torch
sys.path.insert(0, "/root/repo")
from adafocus_amd import synth, _lib
from adafocus_amd.mobilenet import mobilenet_v2
dev = torch.device("cuda:0")
net = mobilenet_v2().eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 3).items()})
net = net.to(dev)
for n in (1024, 512):
    x4 = torch.randn((n, 224, 224, 4), device=dev); x4[..., 3] = 0
    for rnd in range(2):
        for chunk in (512, 1024, 256, 341, 342):
            with _lib.option("mbv2_chunk", chunk), torch.no_grad():
                for _ in range(2): out = net.features_from_nhwc4(x4)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): out = net.features_from_nhwc4(x4)
                e1.record(); torch.cuda.synchronize()
            print("n %d chunk %d: %.3f ms  peak mem %.1f GB" % (n, chunk, e0.elapsed_time(e1) / 5, torch.cuda.max_memory_allocated() / 1e9), flush=True)
