"""EfficientNet as the Focuser's local CNN -- BASELINE.json config 5 ("EfficientNet-B3 local CNN, T=16, P=144, fp16").

The reference has NO implementation of this configuration on any live path: EfficientNet appears only in dead AR-Net
leftovers (STH/ops/models_ada.py:6,69-75 needs the un-vendored `efficientnet_pytorch`; STH/ops/net_flops_table.py:17,29
lists B3 with feature dimension 1536, 1.80 GFLOPs / 12 M parameters) -- SURVEY.md section 8(c): **parity unpinned**.

``EfficientNetLocalCNN`` (args.local_arch = "efficientnet-b3"): the network config 5 names -- MBConv with squeeze-and-excite,
swish, 3x3 / 5x5 depthwise, B3 widths / depths, BN eps 1e-3 -- adafocus_amd/efficientnet.py on csrc/effnet.hip +
csrc/mbconv_whole.hip, checked against oracle/ref_effnet.py (the published algorithm of `efficientnet_pytorch`).  It runs with
activations and 1x1 filters stored as fp16 (fp32 accumulate) or fp32 and is reported by bench.py under `also`, never as `value`.
(Round 2's MobileNetV2-topology stand-in, local_arch = "mbconv_f16", is gone: round 3 built the real network.)
"""
from .efficientnet import EfficientNet

__all__ = ["EfficientNetLocalCNN", "mbconv_local"]


class EfficientNetLocalCNN(EfficientNet):
    """EfficientNet as the Focuser's local CNN: `fc` aliases efficientnet_pytorch's `_fc` (the attribute the reference's
    Focuser replaces on its ResNet, ACT/models/gfv_net.py:262-263); features_nhwc4 / get_featmap come from the base class."""

    def __init__(self, model_name="efficientnet-b3", num_classes=200, dtype="f16", image_size="native"):
        super().__init__(model_name, num_classes=num_classes, image_size=image_size, dtype=dtype)
        self.tsm_segments, self.tsm_div = 0, 8

    @property
    def fc(self):
        return self._fc


def mbconv_local(arch="efficientnet-b3", **kwargs):
    return EfficientNetLocalCNN(arch, **kwargs)
