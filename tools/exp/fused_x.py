"""fused expand + depthwise (dw_same_kernel XN > 0): agreement with the two-launch plan per block boundary, and the network time per fused block."""
import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import synth, _lib as L
from adafocus_amd.efficientnet import EfficientNet
from adafocus_amd.utils import nchw_to_nhwc4
dev = torch.device('cuda:0')
torch.manual_seed(11)
m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype='f16').eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
x4 = nchw_to_nhwc4(torch.randn(5, 3, 144, 144, device=dev) * 0.5)
with torch.no_grad():
    m.features_nhwc4(x4)
    net = m._net
    for bi in range(2, 10):
        with L.option("effnet_plan", 31):
            ref = net.forward_blocks(x4, bi).float().clone()
        with L.option("effnet_plan", 63):
            got = net.forward_blocks(x4, bi).float().clone()
        d = (got - ref).abs()
        print("after block %d %s: max |d| %.3e (max |ref| %.2f), %d of %d differ" % (bi - 1, tuple(ref.shape), float(d.max()), float(ref.abs().max()), int((d > 0).sum()), d.numel()), flush=True)
    if len(sys.argv) > 1:
        n = int(sys.argv[1])
        xb = torch.randn((n, 144, 144, 4), device=dev); xb[..., 3] = 0
        def t(mask):
            with L.option("effnet_plan", 63), L.option("effnet_fused_blocks", mask):
                for _ in range(2): m.features_nhwc4(xb)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): m.features_nhwc4(xb)
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 10
        base = t(0)
        print("none fused: %.3f ms" % base)
        for b in range(2, 9):
            print("block %d fused: %+.3f ms" % (b, t(1 << b) - base), flush=True)
        print("all fused: %.3f ms" % t(0xffffffff))
