"""MBConv local CNN in half-precision storage -- BASELINE.json config 5 ("EfficientNet-B3 local CNN, T=16, P=144, fp16").

The reference has NO implementation of this configuration on any live path: EfficientNet appears only in dead AR-Net
leftovers (STH/ops/models_ada.py:6,69-75 needs the un-vendored `efficientnet_pytorch`; STH/ops/net_flops_table.py:25-30
lists B3 at 1.80 GFLOPs / 12 M parameters) -- SURVEY.md section 8(c): **parity unpinned**.  What this module provides is
the MBConv workload the north-star names, assembled from the network the reference DOES ship with inverted-residual
(MBConv without squeeze-excite) blocks -- its MobileNetV2 (ACT/models/mobilenet.py:42-148) -- run as the local CNN on
patches, with activations and 1x1 weights stored as fp16 (fp32 accumulate; csrc/conv_gemm.hip DT variants,
csrc/misc_ops.hip depthwise).  Its fp32 form is pinned by the G5 golden; the fp16 form is checked against that golden at
fp16 tolerance.  It is reported by bench.py under `also`, never as `value`.
"""
from torch import nn

from .mobilenet import MobileNetV2

__all__ = ["MBConvLocalCNN", "mbconv_local"]


class MBConvLocalCNN(nn.Module):
    """Same surface as adafocus_amd.resnet.ResNet where the Focuser uses it: features_nhwc4(patches, out=) -> (N, 1280)."""

    def __init__(self, num_classes=200, dtype="f16"):
        super().__init__()
        self.net = MobileNetV2(num_classes=num_classes)
        self.net._engine.dtype = dtype
        self.fc = self.net.classifier[-1]
        self.tsm_segments, self.tsm_div = 0, 8

    def features_nhwc4(self, patches_nhwc4, out=None):
        if self.training:
            raise RuntimeError("adafocus_amd.MBConvLocalCNN implements the eval-mode (offline inference) path only")
        _, fvec = self.net.features_from_nhwc4(patches_nhwc4)
        if out is not None:
            from . import hip_ops
            hip_ops.copy2d(fvec, out)
            return out
        return fvec

    def forward(self, x):
        return self.net(x)

    @property
    def feature_dim(self):
        return self.net.last_channel


def mbconv_local(**kwargs):
    return MBConvLocalCNN(**kwargs)
