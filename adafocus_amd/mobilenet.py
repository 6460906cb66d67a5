"""Glancer backbone (MobileNetV2) with the key layout of ACT/models/mobilenet.py.

The nn.Modules below only hold parameters under the reference's names; ``get_featmap`` runs on
``adaf_mobilenetv2`` (expand / project 1x1 convs on the MFMA engine, depthwise 3x3 on the VALU
kernel -- SURVEY.md §8 a10 / f2) and returns ``(featmap, featmap.mean([2,3]))`` like
mobilenet.py:146-148.  ``features_nhwc`` is the layout-native fast path the model composition uses.
"""
from torch import nn

from . import hip_ops
from .glancer_hip import GlancerEngine
from .utils import nchw_to_nhwc4

__all__ = ["MobileNetV2", "mobilenet_v2", "InvertedResidual"]

# (expand t, channels c, repeats n, stride s) -- mobilenet.py:88-97
_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def _cbr(cin, cout, k=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        seq = [] if expand_ratio == 1 else [_cbr(inp, hidden, 1)]
        seq += [_cbr(hidden, hidden, 3, stride, hidden), nn.Conv2d(hidden, oup, 1, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*seq)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.last_channel = 1280
        feats, cin = [_cbr(3, 32, 3, 2)], 32
        for t, c, n, s in _SETTING:
            for i in range(n):
                feats.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(_cbr(cin, self.last_channel, 1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(self.last_channel, num_classes))
        self._engine = GlancerEngine(self, "act")

    def features_nhwc(self, x_nchw):
        """(N,3,S,S) NCHW frames -> (featmap (N,S/32,S/32,1280) NHWC, mean vector (N,1280))."""
        return self._engine.features(nchw_to_nhwc4(x_nchw))

    def features_from_nhwc4(self, frames_nhwc4):
        return self._engine.features(frames_nhwc4)

    def fused_tail(self):
        return bool(self._engine.fused_tail)

    def forward(self, x):
        lin = self.classifier[-1]
        return hip_ops.linear(self.features_nhwc(x)[1], lin.weight.detach(), lin.bias.detach())

    def get_featmap(self, x):
        fmap, fvec = self.features_nhwc(x)
        return fmap.permute(0, 3, 1, 2), fvec

    @property
    def feature_dim(self):
        return self.last_channel


def mobilenet_v2(pretrained=False, progress=True, **kwargs):
    return MobileNetV2(**kwargs)
