import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth
from adafocus_amd.resnet import resnet50
dev = torch.device("cuda:0")
net = resnet50(num_classes=200).eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net = net.to(dev)
trunk = net._sync()
for n in (2, 16, 64):
    x = torch.randn((n, 96, 96, 4), device=dev); x[..., 3] = 0
    out = torch.empty((n, 2048), device=dev)
    for _ in range(5): trunk.forward(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): trunk.forward(x, out=out)
    t1 = time.perf_counter()   # host enqueue time
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("N=%d: host enqueue %.3f ms/forward, end-to-end %.3f ms/forward" % (n, (t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
