import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd import synth
from adafocus_amd.mobilenet import mobilenet_v2
dev = torch.device("cuda:0")
net = mobilenet_v2().eval()
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 3).items()})
net = net.to(dev)
for n, size in ((6, 224), (3, 200), (256, 224)):
    x4 = torch.randn((n, size, size, 4), device=dev); x4[..., 3] = 0
    with torch.no_grad():
        net._engine.fusion = True
        a = [t.clone() for t in net.features_from_nhwc4(x4)]
        net._engine.fusion = False
        b = [t.clone() for t in net.features_from_nhwc4(x4)]
    print(n, size, "equal map:", torch.equal(a[0], b[0]), "vec:", torch.equal(a[1], b[1]), "maxdiff", (a[0]-b[0]).abs().max().item(), "absmax", b[0].abs().max().item())
x4 = torch.randn((1024, 224, 224, 4), device=dev); x4[..., 3] = 0
for fus in (True, False):
    net._engine.fusion = fus
    with torch.no_grad():
        for _ in range(2): net.features_from_nhwc4(x4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): net.features_from_nhwc4(x4)
        e1.record(); torch.cuda.synchronize()
    print("fusion", fus, "glancer 1024 frames: %.3f ms" % (e0.elapsed_time(e1) / 3))
