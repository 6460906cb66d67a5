"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.   **PARITY UNPINNED** (see below).

CPU (PyTorch fp32) restatement of EfficientNet (B0..B7 by compound scaling; B3 is BASELINE.json config 5's local CNN),
written functionally over a flat state dict with the key names of the package the reference names.

Why "unpinned": the reference has NO EfficientNet on any live path.  It appears only in dead AR-Net leftovers:
  * STH/ops/models_ada.py:6     `from efficientnet_pytorch import EfficientNet` (third-party, NOT vendored, no version pin,
                                not installed in this image), used at :69-75 (`EfficientNet.from_pretrained(model_name)` /
                                `EfficientNet.from_named(model_name)` -- the latter does not exist in the package: dead code);
  * STH/ops/net_flops_table.py:17,29  feature dimension 1536 and the prior (1.80 GFLOPs, 12 M parameters) for "efficientnet-b3".
So there is no reference output to compare with.  What IS restated here is the PUBLISHED algorithm of that dependency
(lukemelas/EfficientNet-PyTorch, `efficientnet_pytorch` 0.7.x: model.py `MBConvBlock`, `EfficientNet.extract_features`;
utils.py `round_filters`, `round_repeats`, `Conv2dStaticSamePadding`, `MemoryEfficientSwish`, `efficientnet_params`,
`BlockDecoder`), and the only reference-held numbers are asserted in tests/test_effnet_oracle.py:
feature dimension 1536, ~12 M parameters, 1.8 G multiply-adds at the network's native 300^2 resolution (the reference's
table files the figure under 224 and rescales it by (res/224)^2 -- net_flops_table.py:35-37 -- which is its own
approximation; at 224^2 the published network costs 0.99 G).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this file.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3   # efficientnet_pytorch utils.py efficientnet(): batch_norm_epsilon=1e-3
SE_RATIO = 0.25

# utils.py efficientnet_params(): (width_coefficient, depth_coefficient, native resolution, dropout)
PARAMS = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2), "efficientnet-b1": (1.0, 1.1, 240, 0.2), "efficientnet-b2": (1.1, 1.2, 260, 0.3),
    "efficientnet-b3": (1.2, 1.4, 300, 0.3), "efficientnet-b4": (1.4, 1.8, 380, 0.4), "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5), "efficientnet-b7": (2.0, 3.1, 600, 0.5),
}
# utils.py efficientnet(): blocks_args strings 'r1_k3_s11_e1_i32_o16_se0.25', ... as (repeats, kernel, stride, expand, in, out)
BASE_BLOCKS = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
               (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))


def round_filters(filters, width, divisor=8):
    """utils.py round_filters (depth_divisor 8, min_depth None)."""
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats, depth):
    """utils.py round_repeats."""
    return int(math.ceil(depth * repeats))


def block_list(width, depth):
    """model.py EfficientNet.__init__: one dict per MBConvBlock, in order."""
    out = []
    for r, k, s, e, i, o in BASE_BLOCKS:
        i, o, r = round_filters(i, width), round_filters(o, width), round_repeats(r, depth)
        for j in range(r):
            cin = i if j == 0 else o
            out.append(dict(k=k, stride=s if j == 0 else 1, expand=e, cin=cin, cout=o, hid=cin * e,
                            sq=max(1, int(cin * SE_RATIO))))
    return out


def stem_channels(width):
    return round_filters(32, width)


def head_channels(width):
    return round_filters(1280, width)


def state_dict_shapes(name="efficientnet-b3", num_classes=1000):
    """Key -> shape, the names efficientnet_pytorch gives its parameters and buffers."""
    width, depth = PARAMS[name][:2]
    sh = {}

    def bn(p, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            sh["%s.%s" % (p, leaf)] = (c,)
        sh[p + ".num_batches_tracked"] = ()

    c0 = stem_channels(width)
    sh["_conv_stem.weight"] = (c0, 3, 3, 3)
    bn("_bn0", c0)
    for bi, b in enumerate(block_list(width, depth)):
        p = "_blocks.%d." % bi
        if b["expand"] != 1:
            sh[p + "_expand_conv.weight"] = (b["hid"], b["cin"], 1, 1)
            bn(p + "_bn0", b["hid"])
        sh[p + "_depthwise_conv.weight"] = (b["hid"], 1, b["k"], b["k"])
        bn(p + "_bn1", b["hid"])
        sh[p + "_se_reduce.weight"] = (b["sq"], b["hid"], 1, 1)
        sh[p + "_se_reduce.bias"] = (b["sq"],)
        sh[p + "_se_expand.weight"] = (b["hid"], b["sq"], 1, 1)
        sh[p + "_se_expand.bias"] = (b["hid"],)
        sh[p + "_project_conv.weight"] = (b["cout"], b["hid"], 1, 1)
        bn(p + "_bn2", b["cout"])
    last = block_list(width, depth)[-1]["cout"]
    ch = head_channels(width)
    sh["_conv_head.weight"] = (ch, last, 1, 1)
    bn("_bn1", ch)
    sh["_fc.weight"] = (num_classes, ch)
    sh["_fc.bias"] = (num_classes,)
    return sh


def same_pad(size, k, stride):
    """utils.py Conv2dStaticSamePadding: (pad_before, pad_after) along one axis for an input of `size` (TensorFlow SAME)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + (k - 1) + 1 - size, 0)
    return total // 2, total - total // 2


def out_size(size, stride):
    """utils.py calculate_output_image_size."""
    return -(-size // stride)


def _conv_same(x, w, stride, pad_size, groups=1, bias=None):
    pb, pa = same_pad(pad_size, w.shape[-1], stride)
    if pb or pa:
        x = F.pad(x, (pb, pa, pb, pa))
    return F.conv2d(x, w, bias, stride, 0, 1, groups)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def swish(x):
    """utils.py MemoryEfficientSwish / Swish: x * sigmoid(x)."""
    return x * torch.sigmoid(x)


def mbconv(sd, p, x, b, pad_size):
    """model.py MBConvBlock.forward (eval: drop_connect is the identity).  pad_size = the image size this block's static
    padding was computed for (== the actual input size when the network is built for the input at hand)."""
    inp = x
    if b["expand"] != 1:
        x = swish(_bn(sd, p + "_bn0", F.conv2d(x, sd[p + "_expand_conv.weight"])))
    x = swish(_bn(sd, p + "_bn1", _conv_same(x, sd[p + "_depthwise_conv.weight"], b["stride"], pad_size, groups=b["hid"])))
    s = F.adaptive_avg_pool2d(x, 1)
    s = swish(F.conv2d(s, sd[p + "_se_reduce.weight"], sd[p + "_se_reduce.bias"]))
    s = F.conv2d(s, sd[p + "_se_expand.weight"], sd[p + "_se_expand.bias"])
    x = torch.sigmoid(s) * x
    x = _bn(sd, p + "_bn2", F.conv2d(x, sd[p + "_project_conv.weight"]))
    if b["stride"] == 1 and b["cin"] == b["cout"]:
        x = x + inp
    return x


def extract_features(sd, x, name="efficientnet-b3", image_size="native", upto=None):
    """model.py EfficientNet.extract_features: (N,3,S,S) -> (N, head, s, s).  image_size: the resolution the static SAME
    padding is computed for -- "native" (default) = the model's own resolution, what EfficientNet.from_name(name) bakes into its
    Conv2dStaticSamePadding layers (utils.py get_model_params: global_params.image_size = res); an int =
    from_name(name, image_size=int); None = the package's Conv2dDynamicSamePadding: the input's own size, i.e. exact
    TensorFlow-SAME behaviour.  upto: stop after that many blocks and return the block output (tests)."""
    width, depth = PARAMS[name][:2]
    if image_size == "native":
        image_size = PARAMS[name][2]
    size = int(image_size or x.shape[-1])
    x = swish(_bn(sd, "_bn0", _conv_same(x, sd["_conv_stem.weight"], 2, size)))
    size = out_size(size, 2)
    for bi, b in enumerate(block_list(width, depth)):
        if upto is not None and bi >= upto:
            return x
        x = mbconv(sd, "_blocks.%d." % bi, x, b, size)
        size = out_size(size, b["stride"])
    if upto is not None:
        return x
    return swish(_bn(sd, "_bn1", F.conv2d(x, sd["_conv_head.weight"])))


def mbconv_block(sd, x, name, bi, image_size="native"):
    """Block bi of the network applied to ITS OWN input x (N, cin, h, w) -- tests compare one block's arithmetic in isolation."""
    width, depth, native = PARAMS[name][:3]
    if image_size == "native":
        image_size = native
    size = out_size(int(image_size or 0), 2) if image_size else None
    blocks = block_list(width, depth)
    for b in blocks[:bi]:
        if size is not None:
            size = out_size(size, b["stride"])
    return mbconv(sd, "_blocks.%d." % bi, x, blocks[bi], size if size is not None else x.shape[-1])


def features_pooled(sd, x, name="efficientnet-b3", image_size="native"):
    """extract_features + _avg_pooling + flatten (model.py EfficientNet.forward up to the dropout): (N, head)."""
    return F.adaptive_avg_pool2d(extract_features(sd, x, name, image_size), 1).flatten(1)


def count_macs_params(name="efficientnet-b3", size=None, num_classes=1000):
    """Multiply-adds of the convolutions + fc (the convention of the EfficientNet paper's 'FLOPS' column) and parameter count."""
    width, depth, native, _ = PARAMS[name]
    size = size or native
    macs = 0
    c0 = stem_channels(width)
    s = out_size(size, 2)
    macs += s * s * c0 * 27
    for b in block_list(width, depth):
        if b["expand"] != 1:
            macs += s * s * b["cin"] * b["hid"]
        s = out_size(s, b["stride"])
        macs += s * s * b["hid"] * b["k"] ** 2
        macs += 2 * b["hid"] * b["sq"]
        macs += s * s * b["hid"] * b["cout"]
    ch = head_channels(width)
    macs += s * s * block_list(width, depth)[-1]["cout"] * ch + ch * num_classes
    params = sum(int(torch.tensor(v).prod()) if v else 0 for k, v in state_dict_shapes(name, num_classes).items()
                 if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    return macs, params
