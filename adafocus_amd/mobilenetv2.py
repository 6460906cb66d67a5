"""Glancer backbone for Something-Something: MobileNetV2 with the key layout of
STH/models/mobilenetv2.py (flat Sequentials).  The modules hold parameters; ``get_featmap`` runs on
``adaf_mobilenetv2`` (temporal shift fused into the expand conv of the residual blocks when
``tsm_segments`` is set by the Glancer) and returns ``(featmap, classifier(mean))`` as
mobilenetv2.py:116-121."""
from torch import nn

from . import hip_ops
from .glancer_hip import GlancerEngine
from .utils import nchw_to_nhwc4

__all__ = ["MobileNetV2", "mobilenet_v2", "InvertedResidual"]

_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(inp * expand_ratio)
        self.stride = stride
        self.use_res_connect = stride == 1 and inp == oup
        dw = [nn.Conv2d(hidden, hidden, 3, stride, 1, groups=hidden, bias=False), nn.BatchNorm2d(hidden),
              nn.ReLU6(inplace=True), nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        pw = [] if expand_ratio == 1 else [nn.Conv2d(inp, hidden, 1, 1, 0, bias=False), nn.BatchNorm2d(hidden),
                                            nn.ReLU6(inplace=True)]
        self.conv = nn.Sequential(*(pw + dw))

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class MobileNetV2(nn.Module):
    def __init__(self, n_class=1000, input_size=224, width_mult=1.0):
        super().__init__()
        self.last_channel = 1280
        feats = [nn.Sequential(nn.Conv2d(3, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU6(inplace=True))]
        cin = 32
        for t, c, n, s in _SETTING:
            for i in range(n):
                feats.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        feats.append(nn.Sequential(nn.Conv2d(cin, 1280, 1, 1, 0, bias=False), nn.BatchNorm2d(1280), nn.ReLU6(inplace=True)))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Linear(1280, n_class)
        self.tsm_segments, self.tsm_div = 0, 8
        self._engine = GlancerEngine(self, "sth")

    def features_nhwc(self, x_nchw):
        return self._engine.features(nchw_to_nhwc4(x_nchw), self.tsm_segments, self.tsm_div)

    def forward(self, x):
        return hip_ops.linear(self.features_nhwc(x)[1], self.classifier.weight.detach(), self.classifier.bias.detach())

    def get_featmap(self, x):
        fmap, fvec = self.features_nhwc(x)
        return fmap.permute(0, 3, 1, 2), hip_ops.linear(fvec, self.classifier.weight.detach(), self.classifier.bias.detach())

    @property
    def feature_dim(self):
        return self.last_channel


def mobilenet_v2(n_class, pretrained=False):
    return MobileNetV2(n_class=n_class)
