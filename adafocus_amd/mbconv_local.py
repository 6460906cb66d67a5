"""MBConv local CNNs -- BASELINE.json config 5 ("EfficientNet-B3 local CNN, T=16, P=144, fp16").

The reference has NO implementation of this configuration on any live path: EfficientNet appears only in dead AR-Net
leftovers (STH/ops/models_ada.py:6,69-75 needs the un-vendored `efficientnet_pytorch`; STH/ops/net_flops_table.py:17,29
lists B3 with feature dimension 1536, 1.80 GFLOPs / 12 M parameters) -- SURVEY.md section 8(c): **parity unpinned**.

* ``EfficientNetLocalCNN`` (local_arch = "efficientnet-b3", round 3): the network config 5 names -- MBConv with
  squeeze-and-excite, swish, 3x3 / 5x5 depthwise, B3 widths / depths, BN eps 1e-3 -- in adafocus_amd/efficientnet.py on
  csrc/effnet.hip, checked against oracle/ref_effnet.py (the published algorithm of `efficientnet_pytorch`).
* ``MBConvLocalCNN`` (local_arch = "mbconv_f16" / "mbconv_f32", round 2): the reference's own inverted-residual network
  (its MobileNetV2, ACT/models/mobilenet.py:42-148: MBConv WITHOUT squeeze-excite) as a stand-in; its fp32 form is pinned
  by the G5 golden.  Kept because it is the only MBConv network with a reference-generated golden.

Both run with activations and 1x1 filters stored as fp16 (fp32 accumulate) or fp32.  They are reported by bench.py under
`also`, never as `value`.
"""
from torch import nn

from .efficientnet import EfficientNet
from .mobilenet import MobileNetV2

__all__ = ["MBConvLocalCNN", "EfficientNetLocalCNN", "mbconv_local"]


class MBConvLocalCNN(nn.Module):
    """Same surface as adafocus_amd.resnet.ResNet where the Focuser uses it: features_nhwc4(patches, out=) -> (N, 1280)."""

    def __init__(self, num_classes=200, dtype="f16"):
        super().__init__()
        self.net = MobileNetV2(num_classes=num_classes)
        self.net._engine.dtype = dtype
        self.tsm_segments, self.tsm_div = 0, 8

    @property
    def fc(self):
        # (a property, not a second registration: state_dict() lists the classifier once, under net.classifier.1.*)
        return self.net.classifier[-1]

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


class EfficientNetLocalCNN(EfficientNet):
    """EfficientNet as the Focuser's local CNN: `fc` aliases efficientnet_pytorch's `_fc` (the attribute the reference's
    Focuser replaces on its ResNet, ACT/models/gfv_net.py:262-263); features_nhwc4 / get_featmap come from the base class."""

    def __init__(self, model_name="efficientnet-b3", num_classes=200, dtype="f16", image_size=None):
        super().__init__(model_name, num_classes=num_classes, image_size=image_size, dtype=dtype)
        self.tsm_segments, self.tsm_div = 0, 8

    @property
    def fc(self):
        return self._fc


def mbconv_local(arch="efficientnet-b3", **kwargs):
    if arch.startswith("efficientnet"):
        return EfficientNetLocalCNN(arch, **kwargs)
    return MBConvLocalCNN(**kwargs)
