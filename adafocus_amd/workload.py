"""Algorithmic work of the hot path's pieces: the figures `bench.py`'s roofline
entries are priced with.  Pure arithmetic on layer tables -- no tensors, no device.

"Algorithmic" = what the computation needs if every tensor crosses HBM exactly once per kernel that
must see it: FLOP = 2 x MAC over the real (un-padded) channels; bytes = inputs + outputs (+ residual)
+ weights of each LAUNCH of the plan that actually runs, so a fused plan is priced against fused bytes.
"""

__all__ = ["resnet50_macs_per_patch", "crop_bytes_per_patch", "gru_cls_macs_per_clip", "mobilenetv2_bytes_per_frame",
           "mobilenetv2_macs_per_frame", "hot_path_flops_per_clip", "effnet_blocks", "effnet_macs_per_frame",
           "effnet_bytes_per_frame"]

_STAGE_BLOCKS = (3, 4, 6, 3)
_STAGE_PLANES = (64, 128, 256, 512)
# (t, c, n, s) of ACT/models/mobilenet.py:85-93 (= STH/models/mobilenetv2.py:75-83)
_MBV2 = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def _out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def resnet50_macs_per_patch(patch):
    """MACs of ResNet.get_featmap (ACT/models/resnet.py:211-225) for one patch x patch input:
    0.7507 G at 96, 1.3346 G at 128."""
    hw = _out(patch, 7, 2, 3)
    macs = hw * hw * 64 * 147
    hw = _out(hw, 3, 2, 1)
    inplanes = 64
    for s, (nb, planes) in enumerate(zip(_STAGE_BLOCKS, _STAGE_PLANES)):
        for b in range(nb):
            stride = 2 if (b == 0 and s > 0) else 1
            ohw = _out(hw, 3, stride, 1)
            macs += hw * hw * inplanes * planes                 # conv1 1x1
            macs += ohw * ohw * planes * planes * 9             # conv2 3x3 (stride here, resnet.py:86)
            macs += ohw * ohw * planes * planes * 4             # conv3 1x1
            if b == 0:
                macs += ohw * ohw * inplanes * planes * 4       # downsample 1x1 (+stride)
            inplanes = planes * 4
            hw = ohw
    return macs


def crop_bytes_per_patch(patch, channels=3, elem=4):
    """Window read + patch write: 221 184 B at P = 96 fp32."""
    return 2 * channels * patch * patch * elem


def gru_cls_macs_per_clip(steps, feat=3328, hidden=1024, classes=200):
    """RecurrentClassifier (ACT/models/gfv_net.py:413-435): 3 gates x (input + recurrent) + FC per step."""
    return steps * (3 * hidden * (feat + hidden) + hidden * classes)


def hot_path_flops_per_clip(steps, patch):
    return 2.0 * (steps * resnet50_macs_per_patch(patch) + gru_cls_macs_per_clip(steps))


def _mbv2_blocks(size):
    hw = _out(size, 3, 2, 1)
    cin = 32
    for t, c, n, s in _MBV2:
        for i in range(n):
            stride = s if i == 0 else 1
            yield dict(inp=cin, oup=c, t=t, stride=stride, hw=hw, ohw=_out(hw, 3, stride, 1))
            hw = _out(hw, 3, stride, 1)
            cin = c


def mobilenetv2_macs_per_frame(size=224):
    hw = _out(size, 3, 2, 1)
    macs = hw * hw * 32 * 27
    last = None
    for b in _mbv2_blocks(size):
        hid = b["inp"] * b["t"]
        if b["t"] != 1:
            macs += b["hw"] ** 2 * b["inp"] * hid
        macs += b["ohw"] ** 2 * hid * 9
        macs += b["ohw"] ** 2 * hid * b["oup"]
        last = b
    macs += last["ohw"] ** 2 * last["oup"] * 1280
    return macs


def mobilenetv2_bytes_per_frame(size=224, fused=True, fused_tail=False, whole_blocks=True, strips=True):
    """HBM bytes per frame of the glancer's launch plan (csrc/mobilenetv2.hip): every launch's activation inputs +
    outputs (+ residual), fp32, weights ignored (2.2 M parameters shared by >= 512 frames per launch).
    fused=True: stem + block 1 are one kernel and the expand -> depthwise pairs of the blocks with cin <= 32 on maps
    >= 28^2 are one kernel each (csrc/mbconv.hip), so their wide intermediates never reach HBM.
    fused_tail=True: additionally expand -> depthwise -> project of the 14^2 / 7^2 blocks are one kernel each.
    whole_blocks=True (with fused): the stride-1 blocks with cin, cout <= 32 on maps >= 28^2 (b3, b5, b6) run expand ->
    depthwise -> project + identity in one kernel (mb_block_w_kernel): block input in (and once more as the identity), output out.
    strips=True (round 6, csrc/mbstrip.hip; maps whose output side is a multiple of 14): the stride-2 blocks with 16 / 24 input channels
    (b2, b4) are whole-block launches too -- their depthwise map and the project launch's read of it are gone -- and the stride-1 blocks with
    64 / 96 input channels (b8-b13) run expand -> depthwise in one launch (no expanded map in HBM)."""
    hw = _out(size, 3, 2, 1)
    elems = size * size * 4                                  # pixel-major NHWC4 frames, read once
    first = True
    last = None
    for b in _mbv2_blocks(size):
        hid = b["inp"] * b["t"]
        hin, hout = b["hw"] ** 2, b["ohw"] ** 2
        res = b["stride"] == 1 and b["inp"] == b["oup"]
        if first:
            first = False
            if fused:
                elems += hout * b["oup"]                     # stem + dw + project in one kernel: only its output is written
            else:
                elems += hw * hw * 32 + (hin * hid + hout * hid) + (hout * hid + hout * b["oup"])
            last = b
            continue
        pair_fused = fused and b["inp"] <= 32 and b["hw"] >= 28
        # round 6: expand -> depthwise on strips for the stride-1 blocks with 64 / 96 input channels on maps whose side is a multiple of 14 (b8-b13 at 224^2)
        if fused and whole_blocks and strips and b["stride"] == 1 and b["inp"] in (64, 96) and b["hw"] % 14 == 0 and hid <= 576:
            pair_fused = True
        # ... and for the stride-2 block with 96 (b14: 14^2 -> 7^2)
        if fused and whole_blocks and strips and b["stride"] == 2 and b["inp"] == 96 and b["hw"] % 14 == 0 and hid <= 576:
            pair_fused = True
        s2_whole = strips and b["stride"] == 2 and b["inp"] in (16, 24) and b["ohw"] % 14 == 0
        if pair_fused and whole_blocks and (b["stride"] == 1 or s2_whole) and b["oup"] <= 32 and hid <= 192:
            elems += hin * b["inp"] + hout * b["oup"] + (hout * b["oup"] if res else 0)
        elif fused_tail and b["hw"] <= 14:
            elems += hin * b["inp"] + hout * b["oup"]        # whole block in one kernel (residual = its own input)
        else:
            if pair_fused:
                elems += hin * b["inp"] + hout * hid
            else:
                elems += (hin * b["inp"] + hin * hid) + (hin * hid + hout * hid)
            elems += hout * hid + hout * b["oup"] + (hout * b["oup"] if res else 0)
        last = b
    ohw = last["ohw"] ** 2
    elems += ohw * last["oup"] + ohw * 1280                   # head 1x1
    elems += ohw * 1280 + 1280                                # mean-pooled vector
    return 4 * elems


def mobilenetv2_block_bytes_per_frame(size=224, elem=4):
    """BLOCK-LEVEL algorithmic bytes per frame of the glancer (the denominator config 5 uses, effnet_block_bytes_per_frame): the frame
    (3 channels, fp32) read once, every tensor that crosses a block boundary (stem -> b1 -> ... -> b17 -> head) written once and read
    once in the storage type, the identity re-read where a block has one, the 1280-channel map written (it is the policy's input) and
    the mean-pooled vector written (fp32); nothing inside a block (expanded map, depthwise output) and no parameters.  9.1 MB for 224^2
    frames in fp32 storage -- what a perfectly fused network would move (5.2 MB in fp16 storage); the launch plan that runs moves 22.2 MB
    (mobilenetv2_bytes_per_frame) and the counters see 25.9 MB (profiles/r4_glancer_traffic.json)."""
    hw = _out(size, 3, 2, 1)
    b_ = size * size * 3 * 4 + hw * hw * 32 * elem                      # frame in, stem out
    last = None
    for b in _mbv2_blocks(size):
        b_ += (b["hw"] ** 2 * b["inp"] + b["ohw"] ** 2 * b["oup"]) * elem
        if b["stride"] == 1 and b["inp"] == b["oup"]:
            b_ += b["ohw"] ** 2 * b["oup"] * elem                          # identity rows
        last = b
    ohw = last["ohw"] ** 2
    b_ += ohw * last["oup"] * elem + ohw * 1280 * 4 + 1280 * 4             # head in, map out (fp32: the policy reads it), pooled vector
    return b_


# ---- EfficientNet (BASELINE config 5; csrc/effnet.hip) -----------------------------------------------------------------
# efficientnet_pytorch utils.py: (width, depth) per model; blocks_args (repeats, kernel, stride, expand, in, out)
_EFF_PARAMS = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1), "efficientnet-b2": (1.1, 1.2), "efficientnet-b3": (1.2, 1.4),
               "efficientnet-b4": (1.4, 1.8), "efficientnet-b5": (1.6, 2.2), "efficientnet-b6": (1.8, 2.6), "efficientnet-b7": (2.0, 3.1)}
_EFF_BLOCKS = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
               (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))


def _eff_round(filters, width):
    filters *= width
    new = max(8, int(filters + 4) // 8 * 8)
    return int(new + 8 if new < 0.9 * filters else new)


def effnet_blocks(name, size):
    """(stem channels, [block dicts with the map sizes at `size` input], head channels); SAME padding: out = ceil(in / stride)."""
    import math
    width, depth = _EFF_PARAMS[name]
    hw = -(-size // 2)
    blocks = []
    for r, k, s, e, i, o in _EFF_BLOCKS:
        i, o, r = _eff_round(i, width), _eff_round(o, width), int(math.ceil(depth * r))
        for j in range(r):
            cin, stride = (i if j == 0 else o), (s if j == 0 else 1)
            ohw = -(-hw // stride)
            blocks.append(dict(k=k, stride=stride, expand=e, cin=cin, cout=o, hid=cin * e, sq=max(1, int(cin * 0.25)), hw=hw, ohw=ohw))
            hw = ohw
    return _eff_round(32, width), blocks, _eff_round(1280, width)


def effnet_macs_per_frame(name="efficientnet-b3", size=144):
    """Multiply-adds of extract_features (+ nothing for the pooling): 0.432 G for B3 at 144^2, 1.83 G at its native 300^2."""
    c0, blocks, ch = effnet_blocks(name, size)
    hw = -(-size // 2)
    macs = hw * hw * c0 * 27
    for b in blocks:
        if b["expand"] != 1:
            macs += b["hw"] ** 2 * b["cin"] * b["hid"]
        macs += b["ohw"] ** 2 * b["hid"] * b["k"] ** 2 + 2 * b["hid"] * b["sq"] + b["ohw"] ** 2 * b["hid"] * b["cout"]
    return macs + blocks[-1]["ohw"] ** 2 * blocks[-1]["cout"] * ch


def effnet_whole_block(b, elem=2):
    """Does block `b` (a dict of effnet_blocks) run as ONE launch (csrc/mbconv_whole.hip)?  Mirror of plan_mbw / adaf_mbw_eligible:
    fp16 storage, an expand conv, stride 1, a 3 x 3 ... 9 x 9 map, hidden <= 2048 channels, and LDS for the block input (k padded to
    16) + the depthwise output of one image (two up to 5 x 5) within 160 KB."""
    if elem != 2 or b["expand"] == 1 or b["stride"] != 1 or not (3 <= b["hw"] <= 9) or b["cin"] % 8 or b["hid"] % 16 or b["hid"] > 2048:
        return False
    px, ks = b["hw"] ** 2, -(-b["cin"] // 16)
    for g in ((2, 1) if b["hw"] <= 5 else (1,)):
        for pad in (16, 0):
            first = max(g * px * (ks * 32 + pad), g * (b["hid"] + b["sq"]) * 4)
            if (first + 15) // 16 * 16 + g * px * (b["hid"] * 2 + pad) <= 160 * 1024:
                return True
    return False


def effnet_fused_expand_block(b, elem=2):
    """Does block `b` compute its expand conv inside the depthwise launch (dw_same_kernel XN > 0, csrc/effnet.hip)?  Mirror of
    adaf_launch_dw_expand: fp16 storage, an expand conv, cin a multiple of 8 and at most 64, a channel slice of 48 or 64, and not a
    whole-image block."""
    if elem != 2 or b["expand"] == 1 or effnet_whole_block(b, elem) or b["cin"] % 8 or b["cin"] > 64:
        return False
    return effnet_fused_expand_slices(b) > 0


def effnet_fused_expand_slices(b):
    """Channel slices of the fused launch (each slice's workgroups read the block input again; L2 absorbs most of it): hid / (8 LPP) with
    LPP the largest divisor <= 8 of hid / 8 (plan_dw); 0 when the slice is not 48 or 64 channels."""
    chunks = b["hid"] // 8
    lpp = next(d for d in range(8, 0, -1) if chunks % d == 0)
    return b["hid"] // (8 * lpp) if 8 * lpp in (48, 64) else 0


def effnet_block_bytes_per_frame(name="efficientnet-b3", size=144, elem=2):
    """BLOCK-LEVEL algorithmic bytes per frame: the patch (3 channels, fp32) read once, every tensor that crosses a block boundary
    (stem -> block 0 -> ... -> block 25 -> head) written once and read once in the storage type, the identity skip re-read where a
    block has one, the pooled feature vector written (fp32); nothing inside a block (expanded map, depthwise output, squeeze, gate)
    and no parameters.  3.4 MB for B3 at 144^2 in fp16 storage: the floor a perfectly fused network would move."""
    c0, blocks, ch = effnet_blocks(name, size)
    hw = -(-size // 2)
    b_ = size * size * 3 * 4 + hw * hw * c0 * elem                 # patch in, stem out
    for b in blocks:
        b_ += (b["hw"] ** 2 * b["cin"] + b["ohw"] ** 2 * b["cout"]) * elem
        if b["stride"] == 1 and b["cin"] == b["cout"]:
            b_ += b["ohw"] ** 2 * b["cout"] * elem                    # identity rows
    b_ += blocks[-1]["ohw"] ** 2 * blocks[-1]["cout"] * elem + ch * 4    # head in, pooled vector out
    return b_


def effnet_structural_bytes_per_frame(name="efficientnet-b3", size=144, elem=2, lds_bytes=160 * 1024):
    """A REACHABLE floor for a squeeze-and-excite network (round 6): the block-level bytes of
    effnet_block_bytes_per_frame + the round trip of the depthwise output of every block whose depthwise map of ONE image does not fit
    one CU's LDS (ohw^2 x hid x elem > 160 KB).  The gate multiplies the depthwise map by a function of its global average, so the
    project conv cannot start before the whole map exists: where the map cannot wait on chip it has to be written and read back once.
    That is structural, whatever the kernels do; the block-level figure (4.06 MB per 144^2 patch in fp16 storage) is the floor of a
    network WITHOUT the gate.  B3 at 144^2, fp16 storage: blocks 0-8 (72^2 ... 18^2 maps) exceed the LDS."""
    c0, blocks, ch = effnet_blocks(name, size)
    b_ = effnet_block_bytes_per_frame(name, size, elem)
    for b in blocks:
        if b["ohw"] ** 2 * b["hid"] * elem > lds_bytes:
            b_ += 2 * b["ohw"] ** 2 * b["hid"] * elem
    return b_


def effnet_bytes_per_frame(name="efficientnet-b3", size=144, elem=2, fused=True):
    """HBM bytes per frame of the launch plan of csrc/effnet.hip + csrc/mbconv_whole.hip that RUNS (activation inputs + outputs of every
    launch; the 12 M parameters are shared by >= 1024 frames per launch and ignored): patch in (fp32 x 4 lanes), stem out; per block
    either ONE launch (block in, out, + identity: the whole-image kernel, fused=True and effnet_whole_block), or expand + depthwise in one
    launch (block in once per channel slice, depthwise out: effnet_fused_expand_block) + project, or expand (in, out), depthwise (in, out),
    project (in, out, + identity); head (in, fp32 out), pooled vector.  elem = bytes per stored activation
    (2 = fp16 storage, 4 = fp32)."""
    c0, blocks, ch = effnet_blocks(name, size)
    hw = -(-size // 2)
    b_ = size * size * 16 + hw * hw * c0 * elem
    for b in blocks:
        hin, hout = b["hw"] ** 2, b["ohw"] ** 2
        if fused and effnet_whole_block(b, elem):
            b_ += (hin * b["cin"] + hout * b["cout"]) * elem
            if b["cin"] == b["cout"]:
                b_ += hout * b["cout"] * elem
            continue
        if fused and effnet_fused_expand_block(b, elem):
            b_ += (hin * b["cin"] * effnet_fused_expand_slices(b) + hout * b["hid"]) * elem      # expand + depthwise in one launch: narrow input (once per channel slice), depthwise output
        else:
            if b["expand"] != 1:
                b_ += (hin * b["cin"] + hin * b["hid"]) * elem
            b_ += (hin * b["hid"] + hout * b["hid"]) * elem                   # depthwise
        b_ += (hout * b["hid"] + hout * b["cout"]) * elem                      # project
        if b["stride"] == 1 and b["cin"] == b["cout"]:
            b_ += hout * b["cout"] * elem
    last = blocks[-1]
    b_ += last["ohw"] ** 2 * (last["cout"] * elem + ch * 4) + last["ohw"] ** 2 * ch * 4 + ch * 4
    return b_
