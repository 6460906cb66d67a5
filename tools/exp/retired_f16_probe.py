import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from adafocus_amd import synth
from adafocus_amd.mobilenet import mobilenet_v2
from tests.helpers import golden, rnd
dev = torch.device('cuda:0')
g = golden("g5_mbv2_act")
mb = mobilenet_v2().eval()
shapes = {k: tuple(v.shape) for k, v in mb.state_dict().items()}
mb.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 505).items()})
mb = mb.to(dev)
xc = rnd((2, 3, 64, 64), 53).to(dev)
fm32, fv32 = mb.features_nhwc(xc); fm32 = fm32.clone(); fv32 = fv32.clone()
mb._engine.dtype = "f16"
fm16, fv16 = mb.features_nhwc(xc)
ref = torch.from_numpy(g["fm"]).permute(0, 2, 3, 1)
print("scale", float(ref.abs().max()), "f32 err", (fm32.cpu()-ref).abs().max().item(), "f16 max err", (fm16.cpu()-ref).abs().max().item(),
      "fv err", (fv16.cpu()-torch.from_numpy(g["fv"])).abs().max().item(), "rel rms", ((fm16.cpu()-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item())
