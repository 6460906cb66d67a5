import sys, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import synth, _lib as L
from adafocus_amd.efficientnet import EfficientNet
dev = torch.device('cuda:0')
m = EfficientNet.from_name('efficientnet-b3', num_classes=200, dtype='f16').eval()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()})
m = m.to(dev)
n = 1024
xb = torch.randn((n, 144, 144, 4), device=dev); xb[..., 3] = 0
def t(mask):
    with L.option("effnet_fused_blocks", mask):
        for _ in range(2): m.features_nhwc4(xb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): m.features_nhwc4(xb)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
with torch.no_grad():
    base = t(0)
    print("none fused %.3f" % base)
    for kb in (32, 48, 64, 80, 100):
        for sc in (24, 48, 96, 200):
            with L.option("dwx_budget_kb", kb), L.option("dwx_stage_cost", sc):
                r = [t(1 << b) - base for b in (2, 3, 5, 6, 8)]
                print("budget %3d KB cost %3d: b2 %+.3f b3 %+.3f b5 %+.3f b6 %+.3f b8 %+.3f  all %.3f" % ((kb, sc) + tuple(r) + (t(0xffffffff),)), flush=True)
