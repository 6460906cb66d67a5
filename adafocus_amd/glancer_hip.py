"""Bridges the two MobileNetV2 key layouts of the reference (ACT/models/mobilenet.py vs
STH/models/mobilenetv2.py) to the layout-neutral parameter names of ``adaf_mobilenetv2`` and keeps
the library's packed copy in sync with the nn.Module's parameters."""
from . import hip_ops

_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
_BN = ("weight", "bias", "running_mean", "running_var")


def _expand_ratios():
    out = []
    for t, c, n, s in _SETTING:
        out += [t] * n
    return out          # for features.1 .. features.17


def neutral_params(state_dict, variant):
    """Reference state dict of the MobileNetV2 module ('features.*') -> {neutral name: tensor}."""
    sd = state_dict
    out = {}

    def conv(dst, key):
        out[dst + ".weight"] = sd[key + ".weight"] if key + ".weight" in sd else sd[key + ".net.weight"]

    def bn(dst, key):
        for leaf in _BN:
            out["%s.bn.%s" % (dst, leaf)] = sd["%s.%s" % (key, leaf)]

    conv("stem", "features.0.0")
    bn("stem", "features.0.1")
    for i, t in enumerate(_expand_ratios(), start=1):
        b, p = "b%d" % i, "features.%d.conv." % i
        if variant == "act":
            k = 0
            if t != 1:
                conv(b + ".expand", p + "0.0")
                bn(b + ".expand", p + "0.1")
                k = 1
            conv(b + ".dw", p + "%d.0" % k)
            bn(b + ".dw", p + "%d.1" % k)
            conv(b + ".project", p + "%d" % (k + 1))
            bn(b + ".project", p + "%d" % (k + 2))
        else:
            k = 0
            if t != 1:
                conv(b + ".expand", p + "0")
                bn(b + ".expand", p + "1")
                k = 3
            conv(b + ".dw", p + "%d" % k)
            bn(b + ".dw", p + "%d" % (k + 1))
            conv(b + ".project", p + "%d" % (k + 3))
            bn(b + ".project", p + "%d" % (k + 4))
    conv("head", "features.18.0")
    bn("head", "features.18.1")
    return out


class GlancerEngine:
    """Lazy HIP twin of a MobileNetV2 nn.Module (parameters stay owned by the module)."""

    def __init__(self, module, variant):
        self.module, self.variant = module, variant
        self._net, self._sig = None, None
        self.fusion = True      # expand -> depthwise fused where the shape allows (adaf_mobilenetv2_set_fusion)
        self.fused_tail = False  # whole inverted-residual blocks of the 14^2 / 7^2 maps in one kernel (not built)

    def sync(self):
        sd = {k: v for k, v in self.module.state_dict().items()
              if k.startswith("features.") and not k.endswith("num_batches_tracked")}
        dev = next(iter(sd.values())).device
        if dev.type != "cuda":
            raise RuntimeError("adafocus_amd glancer runs on MI355X only; move the module to the GPU (.cuda())")
        sig = tuple((v.data_ptr(), v._version) for v in sd.values())
        if self._net is None or self._net.device != dev or sig != self._sig:
            if self._net is None or self._net.device != dev:
                self._net = hip_ops.MobileNetV2Net(dev)
            self._net.load(neutral_params(sd, self.variant))
            self._sig = sig
        self._net.set_fusion(self.fusion)
        return self._net

    def features(self, frames_nhwc4, tsm_segments=0, tsm_div=8, want_vec=True):
        if self.module.training:
            raise RuntimeError("adafocus_amd glancer: eval mode only")
        return self.sync().forward(frames_nhwc4, tsm_segments, tsm_div, want_vec)
