"""EfficientNet with the module / state-dict layout of `efficientnet_pytorch` -- the package the reference names for
BASELINE config 5's local CNN (STH/ops/models_ada.py:6,69-75: `from efficientnet_pytorch import EfficientNet`,
`EfficientNet.from_pretrained(model_name)`; feature dimension 1536 and the (1.80 GFLOPs, 12 M) prior for "efficientnet-b3"
in STH/ops/net_flops_table.py:17,29).  That import is dead AR-Net code, the package is neither vendored nor pinned nor
installed here: **parity unpinned** -- what is mirrored is the package's published interface (`from_name`,
`extract_features`, `forward`, the `_conv_stem` / `_blocks.N._depthwise_conv` / `_se_reduce` ... key names) so a checkpoint
saved from it loads with strict=True, and its published algorithm (oracle/ref_effnet.py restates it on the CPU).

The nn.Modules below only hold parameters; every forward runs on `adaf_effnet` (csrc/effnet.hip).
"""
import math

from torch import nn

from . import hip_ops
from .utils import nchw_to_nhwc4

__all__ = ["EfficientNet", "MBConvBlock", "efficientnet_params"]

# utils.py efficientnet_params(): width, depth, native resolution, dropout
_PARAMS = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2), "efficientnet-b1": (1.0, 1.1, 240, 0.2), "efficientnet-b2": (1.1, 1.2, 260, 0.3),
    "efficientnet-b3": (1.2, 1.4, 300, 0.3), "efficientnet-b4": (1.4, 1.8, 380, 0.4), "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5), "efficientnet-b7": (2.0, 3.1, 600, 0.5),
}
# utils.py efficientnet(): (repeats, kernel, stride, expand ratio, input filters, output filters), se_ratio 0.25
_BLOCKS = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
           (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))
_BN = ("weight", "bias", "running_mean", "running_var")


def efficientnet_params(model_name):
    return _PARAMS[model_name]


def _round_filters(filters, width, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def _round_repeats(repeats, depth):
    return int(math.ceil(depth * repeats))


def _bn(c):
    return nn.BatchNorm2d(c, momentum=0.01, eps=1e-3)


class MBConvBlock(nn.Module):
    """Parameter container with model.py MBConvBlock's attribute names."""

    def __init__(self, cin, cout, k, stride, expand, se_ratio=0.25):
        super().__init__()
        hid = cin * expand
        self.cin, self.cout, self.k, self.stride, self.expand_ratio, self.hid = cin, cout, k, stride, expand, hid
        if expand != 1:
            self._expand_conv = nn.Conv2d(cin, hid, 1, bias=False)
            self._bn0 = _bn(hid)
        self._depthwise_conv = nn.Conv2d(hid, hid, k, stride, groups=hid, bias=False)
        self._bn1 = _bn(hid)
        sq = max(1, int(cin * se_ratio))
        self._se_reduce = nn.Conv2d(hid, sq, 1)
        self._se_expand = nn.Conv2d(sq, hid, 1)
        self._project_conv = nn.Conv2d(hid, cout, 1, bias=False)
        self._bn2 = _bn(cout)

    def forward(self, inputs, drop_connect_rate=None):
        raise NotImplementedError("adafocus_amd runs the whole network through EfficientNet.extract_features (csrc/effnet.hip)")


class EfficientNet(nn.Module):
    def __init__(self, model_name="efficientnet-b3", num_classes=1000, image_size="native", dtype="f32"):
        super().__init__()
        width, depth, native, dropout = _PARAMS[model_name]
        self.model_name, self.width, self.depth = model_name, width, depth
        # the resolution the SAME padding is computed for.  "native" (default) = the model's own resolution, 300 for B3: what
        # the package's EfficientNet.from_name(name) bakes into its Conv2dStaticSamePadding layers (utils.py get_model_params ->
        # global_params.image_size = res), so a checkpoint produced with the package sees the same sampling grid here; an int =
        # from_name(name, image_size=int); None = the package's dynamic form (image_size=None -> Conv2dDynamicSamePadding):
        # padding for the input at hand, i.e. exact TensorFlow-SAME behaviour
        self.image_size = native if image_size == "native" else image_size
        c0 = _round_filters(32, width)
        self._conv_stem = nn.Conv2d(3, c0, 3, 2, bias=False)
        self._bn0 = _bn(c0)
        blocks = []
        for r, k, s, e, i, o in _BLOCKS:
            i, o, r = _round_filters(i, width), _round_filters(o, width), _round_repeats(r, depth)
            for j in range(r):
                blocks.append(MBConvBlock(i if j == 0 else o, o, k, s if j == 0 else 1, e))
        self._blocks = nn.ModuleList(blocks)
        ch = _round_filters(1280, width)
        self._conv_head = nn.Conv2d(blocks[-1].cout, ch, 1, bias=False)
        self._bn1 = _bn(ch)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self._dropout = nn.Dropout(dropout)
        self._fc = nn.Linear(ch, num_classes)
        self.last_layer_name = "_fc"            # STH/ops/models_ada.py:73
        self.feature_dim = ch
        self.storage = dtype                    # "f32" | "f16": HBM storage of activations and 1x1 filters
        self.fusion = True                      # fp16 storage: whole-image MBConv kernels for maps up to 9 x 9 (adaf_effnet_set_fusion; csrc/mbconv_whole.hip)
        self._net, self._sig = None, None

    @classmethod
    def from_name(cls, model_name, in_channels=3, **override_params):
        if in_channels != 3:
            raise NotImplementedError("adafocus_amd EfficientNet: RGB input only")
        return cls(model_name, **override_params)

    # ---- engine ---------------------------------------------------------------------------------------------------
    def _neutral(self, sd):
        out = {}

        def conv(dst, key, bn):
            out[dst + ".weight"] = sd[key + ".weight"]
            for leaf in _BN:
                out["%s.bn.%s" % (dst, leaf)] = sd["%s.%s" % (bn, leaf)]
        conv("stem", "_conv_stem", "_bn0")
        for i, b in enumerate(self._blocks):
            p = "_blocks.%d." % i
            if b.expand_ratio != 1:
                conv("b%d.expand" % i, p + "_expand_conv", p + "_bn0")
            conv("b%d.dw" % i, p + "_depthwise_conv", p + "_bn1")
            conv("b%d.project" % i, p + "_project_conv", p + "_bn2")
            for name in ("se_reduce", "se_expand"):
                out["b%d.%s.weight" % (i, name)] = sd[p + "_" + name + ".weight"]
                out["b%d.%s.bias" % (i, name)] = sd[p + "_" + name + ".bias"]
        conv("head", "_conv_head", "_bn1")
        return out

    def engine(self):
        sd = {k: v for k, v in self.state_dict().items() if not k.startswith("_fc.") and not k.endswith("num_batches_tracked")}
        dev = self._conv_stem.weight.device
        if dev.type != "cuda":
            raise RuntimeError("adafocus_amd EfficientNet runs on MI355X only; move the module to the GPU (.cuda())")
        sig = tuple((v.data_ptr(), v._version) for v in sd.values()) + (self.storage,)
        if self._net is None or self._net.device != dev or sig != self._sig:
            if self._net is None or self._net.device != dev:
                self._net = hip_ops.EffNetNet(dev, self.width, self.depth)
            self._net.set_dtype(self.storage)
            self._net.load(self._neutral(sd))
            self._sig = sig
        self._net.set_fusion(self.fusion)
        self._net.pad_size = int(self.image_size or 0)
        return self._net

    def _check_eval(self):
        if self.training:
            raise RuntimeError("adafocus_amd EfficientNet implements the eval-mode (offline inference) path only")

    # ---- efficientnet_pytorch surface -----------------------------------------------------------------------------
    def extract_features(self, inputs):
        """(N,3,S,S) -> (N, C_head, s, s) -- model.py EfficientNet.extract_features."""
        self._check_eval()
        fmap, _ = self.engine().forward(nchw_to_nhwc4(inputs), self.image_size or 0, want_map=True, want_vec=False)
        return fmap.permute(0, 3, 1, 2)

    def forward(self, inputs):
        """model.py EfficientNet.forward: extract_features -> _avg_pooling -> flatten -> (_dropout) -> _fc."""
        return hip_ops.linear(self.features_nhwc4(nchw_to_nhwc4(inputs)), self._fc.weight.detach(), self._fc.bias.detach())

    # ---- layout-native path used by the Focuser -------------------------------------------------------------------
    def features_nhwc4(self, patches_nhwc4, out=None):
        """(N,P,P,4) pixel-major patches -> pooled feature (N, C_head); `out` = a (N, >= C_head) view to write into."""
        self._check_eval()
        _, fvec = self.engine().forward(patches_nhwc4, self.image_size or 0, want_map=False, want_vec=True, out=out)
        return fvec

    def get_featmap(self, x, pooled=True):
        """The local CNN's call in the reference's Focuser (ACT/models/gfv_net.py:329: net.get_featmap(patch, pooled=True))."""
        if pooled:
            return self.features_nhwc4(nchw_to_nhwc4(x)).view(x.shape[0], -1, 1, 1)
        return self.extract_features(x)
