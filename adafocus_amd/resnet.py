"""Local CNN: ResNet-50 with the reference's parameter names, executed by the HIP trunk.

Mirrors the surface of ACT/models/resnet.py that the hot path uses -- ``resnet50()``,
``ResNet.get_featmap(x, pooled)`` (:211-225), ``get_featvec`` (:227-239), ``forward`` (:196-209),
``feature_dim`` (:241-243) and the torchvision state-dict keys -- while the nn.Conv2d / BatchNorm2d
children only hold parameters: their ``forward`` is never called.  All arithmetic runs in
``adaf_resnet50_forward`` (implicit-GEMM MFMA convs with the BN affine, residual add and ReLU in
the epilogue).  Only ResNet-50 is built: the other depths are never instantiated by the reference
drivers.
"""
import torch
from torch import nn

from . import hip_ops
from .utils import nchw_to_nhwc4

__all__ = ["ResNet", "resnet50", "Bottleneck"]

_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))
DEFAULT_MATH = "f32"      # arithmetic of newly built trunks ("f32" | "split_bf16"); ResNet.set_math overrides per model


class Bottleneck(nn.Module):
    """Parameter container for one residual block (ACT/models/resnet.py:74-114)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, with_downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("Bottleneck is a parameter container; the block runs inside adaf_resnet50_forward")


class ResNet(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for i, (planes, blocks, stride) in enumerate(_STAGES, start=1):
            seq = [Bottleneck(inplanes, planes, stride, True)]
            inplanes = planes * 4
            seq += [Bottleneck(inplanes, planes, 1, False) for _ in range(blocks - 1)]
            setattr(self, "layer%d" % i, nn.Sequential(*seq))
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():  # same initialisation family as the reference (resnet.py:152-157)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self.tsm_segments = 0     # > 0: temporal shift before every Bottleneck conv1 (TSM, STH)
        self.tsm_div = 8
        self.tsm_place = "blockres"   # 'blockres': shift inside every Bottleneck conv1; 'block': in front of the whole Bottleneck
        self._trunk = None
        self._sig = None

    # ---- weight hand-off to the library -------------------------------------------------
    def _trunk_params(self):
        out = {}
        for k, v in self.state_dict().items():
            if not (k.startswith("fc.") or k.endswith("num_batches_tracked")):
                out[k] = v
        return out

    def _sync(self):
        params = self._trunk_params()
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("adafocus_amd.ResNet runs on MI355X only; move the module to the GPU (.cuda())")
        sig = tuple((v.data_ptr(), v._version) for v in params.values())
        if self._trunk is None or self._trunk.device != dev or sig != self._sig:
            if self._trunk is None or self._trunk.device != dev:
                self._trunk = hip_ops.ResNet50Trunk(dev)
                self._trunk.set_math(getattr(self, "_math", None) or DEFAULT_MATH)
                self._trunk.set_fusion(getattr(self, "_fusion", True))
            self._trunk.load(params)
            self._sig = sig
        if getattr(self._trunk, "_place", "blockres") != self.tsm_place:
            self._trunk.set_shift_place(self.tsm_place)
            self._trunk._place = self.tsm_place
        return self._trunk

    def set_fusion(self, on):
        """Trunk layer fusion (stage-1 bottleneck tails, stem + max-pool) on / off; bit-identical either way."""
        self._fusion = int(on)
        if self._trunk is not None:
            self._trunk.set_fusion(on)

    def set_math(self, mode):
        """Opt-in arithmetic of the convolutions: "f32" (default, fp32 matrix pipe) or "split_bf16" (fp32 operands
        as three exact bf16 parts on the bf16 matrix pipe, fp32 accumulate); no reference counterpart."""
        self._math = mode
        if self._trunk is not None:
            self._trunk.set_math(mode)

    # ---- reference surface --------------------------------------------------------------
    def features_nhwc4(self, patches_nhwc4, out=None):
        """Fast path used by the Focuser: (N,P,P,4) pixel-major patches -> (N,2048)."""
        if self.training:
            raise RuntimeError("adafocus_amd.ResNet implements the eval-mode (offline inference) path only")
        return self._sync().forward(patches_nhwc4, self.tsm_segments, self.tsm_div, out=out)

    def features_from_frames(self, frames, actions, patch_size, frames_per_action=1, out=None):
        """PatchSampler.sample + get_featmap(pooled=True) in one call (ACT/models/gfv_net.py:325-331): frames (N,3,H,W) or pixel-major
        (N,H,W,4), actions (k * N / frames_per_action, 2) -> (k * N, 2048); the stem gathers its own windows (no patch tensor)."""
        if self.training:
            raise RuntimeError("adafocus_amd.ResNet implements the eval-mode (offline inference) path only")
        return self._sync().forward_frames(frames, actions.to(device=frames.device, dtype=torch.float32), patch_size, frames_per_action, self.tsm_segments,
                                           self.tsm_div, out=out)

    def get_featmap(self, x, pooled=True):
        """ACT/models/resnet.py:211-225: the trunk up to layer4, then the global average pool (pooled=True: (N,2048,1,1), the call the
        Focuser makes) or the map itself (pooled=False: (N,2048,s,s), NCHW like the reference's return value)."""
        if self.training:
            raise RuntimeError("adafocus_amd.ResNet implements the eval-mode (offline inference) path only")
        if not pooled:
            fmap, _ = self._sync().forward_map(nchw_to_nhwc4(x), self.tsm_segments, self.tsm_div)
            return fmap.permute(0, 3, 1, 2)
        feat = self.features_nhwc4(nchw_to_nhwc4(x))
        return feat.view(feat.shape[0], 2048, 1, 1)

    def get_featvec(self, x):
        return self.features_nhwc4(nchw_to_nhwc4(x))

    def forward(self, x):
        return hip_ops.linear(self.get_featvec(x), self.fc.weight.detach(), self.fc.bias.detach())

    @property
    def feature_dim(self):
        return self.fc.weight.shape[-1]


def resnet50(pretrained=False, progress=True, **kwargs):
    """ACT/models/resnet.py:280.  There is no network here: `pretrained` weights must be supplied
    through load_state_dict (the reference's checkpoints load unchanged)."""
    return ResNet(**kwargs)
