"""Glancer backbone (MobileNetV2) with the key layout of ACT/models/mobilenet.py.

Upstream of the named hot path ("next" row f2): in this round it runs as
stock PyTorch-ROCm modules on the GPU and only PRODUCES the policy input and the 1280-d global
feature; it is not a fallback for any HIP kernel.  ``get_featmap`` returns
``(featmap, featmap.mean([2,3]))`` exactly like mobilenet.py:146-148.
"""
from torch import nn

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

    def forward(self, x):
        return self.classifier(self.features(x).mean([2, 3]))

    def get_featmap(self, x):
        x = self.features(x)
        return x, x.mean([2, 3])

    @property
    def feature_dim(self):
        return self.last_channel


def mobilenet_v2(pretrained=False, progress=True, **kwargs):
    return MobileNetV2(**kwargs)
