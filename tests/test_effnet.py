"""GPU tests of the EfficientNet path (BASELINE config 5's local CNN; csrc/effnet.hip) through the C ABI.

PARITY UNPINNED: the reference has no EfficientNet on a live path, so the checker is
oracle/ref_effnet.py -- the published algorithm of the package the reference names -- and plain torch ops for the building
blocks.  fp32 storage must meet the fp32 bar (1e-3); fp16 storage is held to a measured tolerance, stated per test.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adafocus_amd import synth
from tests.helpers import rnd

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from adafocus_amd import hip_ops
    return hip_ops


@pytest.fixture(scope="module")
def R():
    from oracle import ref_effnet
    return ref_effnet


def _smooth(shape, seed):
    """Inputs with structure at every scale (a coarse random field upsampled + noise): pooled features then differ from
    sample to sample by much more than the tolerance, which N(0,1) pixel noise alone would not achieve."""
    n, c, h, w = shape
    coarse = rnd((n, c, 6, 6), seed, 0.8)
    return F.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False) + rnd(shape, seed + 1, 0.5)


def _close(got, ref, tol=TOL):
    """A RELATIVE fp32 bar: |got - ref| < tol x max(1, max |ref| of the SAMPLE) -- scaled per sample by the magnitude of what is
    compared, not an absolute 1e-3: random-weight EfficientNets amplify a few inputs by two orders of magnitude (26 blocks of swish +
    SE + identity with random BN gains), and fp32 rounding scales with that.  Returns (ok, worst relative error)."""
    n = ref.shape[0]
    scale = ref.reshape(n, -1).abs().amax(1).clamp(min=1.0)
    err = (got - ref).reshape(n, -1).abs().amax(1)
    return bool((err < tol * scale).all()), float((err / scale).max())


def _swish(x):
    return x * torch.sigmoid(x)


# ------------------------------------------------------------------------------------ building blocks
@pytest.mark.parametrize("k,stride,size,c", [(3, 1, 72, 40), (3, 1, 72, 24), (3, 2, 72, 144), (3, 1, 36, 192), (5, 2, 36, 192),
                                             (5, 1, 18, 288), (3, 2, 18, 288), (3, 1, 9, 576), (5, 1, 9, 816), (5, 2, 9, 816),
                                             (5, 1, 5, 1392), (3, 1, 5, 2304), (5, 2, 11, 48), (3, 2, 7, 16), (5, 1, 10, 8),
                                             (3, 1, 13, 200), (5, 1, 4, 64), (3, 1, 4, 96), (5, 1, 3, 64), (3, 1, 3, 2304)])
def test_dwconv_same_vs_torch(dev, ops, R, k, stride, size, c):
    """Depthwise k x k with SAME padding + BN affine + swish + squeeze mean, fp32 and fp16 storage, against F.conv2d on the
    explicitly padded input (Conv2dStaticSamePadding)."""
    n = 11 if size <= 9 else 3          # small maps: several images share a block (8 x 48 channels > 256 threads once bit us)
    x = rnd((n, c, size, size), 700 + size + c)
    w = rnd((c, 1, k, k), 701 + c, 0.3)
    scale = rnd((c,), 702, 0.2) + 1.0
    bias = rnd((c,), 703, 0.1)
    pb, pa = R.same_pad(size, k, stride)
    ref = F.conv2d(F.pad(x, (pb, pa, pb, pa)), w, None, stride, 0, 1, c) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ref = _swish(ref)
    wk = ops.pack_dw_weight_kxk(w.to(dev))
    assert torch.equal(wk.cpu(), w.view(c, k * k).t().contiguous())
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out, pool = ops.dwconv_same_bn_act(xh, wk, scale.to(dev), bias.to(dev), k, stride, ops.ACT_SWISH, want_pool=True)
    assert out.shape == (n, -(-size // stride), -(-size // stride), c)
    assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-5
    assert (pool.cpu() - ref.mean([2, 3])).abs().max().item() < 2e-5
    # no squeeze wanted: same map
    assert torch.equal(ops.dwconv_same_bn_act(xh, wk, scale.to(dev), bias.to(dev), k, stride, ops.ACT_SWISH), out)
    if c % 8 == 0:
        x16 = xh.half()
        ref16 = F.conv2d(F.pad(x16.float().cpu().permute(0, 3, 1, 2), (pb, pa, pb, pa)), w, None, stride, 0, 1, c)
        ref16 = _swish(ref16 * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
        o16, p16 = ops.dwconv_same_bn_act(x16, wk, scale.to(dev), bias.to(dev), k, stride, ops.ACT_SWISH, want_pool=True)
        assert o16.dtype == torch.float16
        # products and sums are fp32 on fp16 inputs: only the final store rounds (half an ulp of fp16 = 2^-11 relative)
        assert (o16.float().cpu().permute(0, 3, 1, 2) - ref16).abs().max().item() < 1e-3 * max(1.0, float(ref16.abs().max()))
        assert (p16.cpu() - ref16.mean([2, 3])).abs().max().item() < 2e-5      # the squeeze sums the fp32 values


def test_tiny_map_depthwise_kernel_bit_identical_to_the_staged_one(dev):
    """dw_small_kernel (H = W <= 5: one thread per image x 4 channels, padding taps skipped, thread-local squeeze sums) against
    dw_same_kernel on the same inputs: same sha256 of every output map, fp32 and fp16 storage, 3x3 and 5x5 windows, 3^2 .. 5^2 maps
    (tools/dw_small_ab.py; the two arms differ in one bit of the library option "effnet_plan")."""
    from adafocus_amd import _lib
    from tests.helpers import load_tool
    tool = load_tool("dw_small_ab")
    plan = int(_lib.get_option("effnet_plan"))
    with _lib.option("effnet_plan", plan | _lib.EF_PLAN_TINY_DW):
        small = tool.digests()
    with _lib.option("effnet_plan", plan & ~_lib.EF_PLAN_TINY_DW):
        staged = tool.digests()
    assert len(small) == 24 and len(staged) == 24
    for a, b in zip(small, staged):
        assert a[:2] == b[:2], (a, b)                                  # the maps: bit for bit
        assert abs(float(a[2]) - float(b[2])) <= 1e-3 * max(1.0, abs(float(b[2])))      # the squeeze sums: another summation order


def test_narrow_project_strip_kernel_bit_identical_to_the_tiled_one(dev):
    """ef_nproj_kernel (blocks 0-1: K <= 64, N <= 32, millions of rows) against gated_project_kernel: same sha256 of the block
    outputs in fp32 and fp16 storage, with and without the identity skip, on even and odd maps (tools/nproj_ab.py)."""
    from adafocus_amd import _lib
    from tests.helpers import load_tool
    tool = load_tool("nproj_ab")
    plan = int(_lib.get_option("effnet_plan"))
    with _lib.option("effnet_plan", plan | _lib.EF_PLAN_STRIP_PROJECT):
        strip = tool.digests()
    with _lib.option("effnet_plan", plan & ~_lib.EF_PLAN_STRIP_PROJECT):
        tiled = tool.digests()
    assert len(strip) == 8 and strip == tiled


def test_dwconv_same_rejects_bad_arguments(dev, ops):
    from adafocus_amd._lib import AdafError
    x = torch.zeros((1, 8, 8, 8), device=dev)
    w = torch.zeros((16, 8), device=dev)
    s = torch.ones(8, device=dev)
    with pytest.raises(AdafError):
        ops.dwconv_same_bn_act(x, w, s, s, 4, 1)            # k = 4
    with pytest.raises(AdafError):
        ops.dwconv_same_bn_act(x, w, s, s, 3, 3)            # stride 3
    with pytest.raises(AdafError):
        ops.dwconv_same_bn_act(torch.zeros((1, 8, 8, 6), device=dev), w, s, s, 3, 1)   # channels do not fill 16-byte chunks


@pytest.mark.parametrize("n,c,sq", [(5, 40, 10), (3, 24, 6), (2, 816, 34), (4, 2304, 96), (1, 8, 1)])
def test_se_gate_vs_torch(dev, ops, n, c, sq):
    m = rnd((n, c), 710 + c, 0.5)
    wr, br = rnd((sq, c, 1, 1), 711, 0.2), rnd((sq,), 712, 0.1)
    we, be = rnd((c, sq, 1, 1), 713, 0.3), rnd((c,), 714, 0.1)
    s = F.conv2d(_swish(F.conv2d(m.view(n, c, 1, 1), wr, br)), we, be)
    ref = torch.sigmoid(s).view(n, c)
    got = ops.se_gate(m.to(dev), wr.to(dev), br.to(dev), we.to(dev), be.to(dev))
    assert (got.cpu() - ref).abs().max().item() < 1e-5           # (sums over up to 2304 channels before the sigmoid)


@pytest.mark.parametrize("hw,cin,cout,res", [(25, 2304, 384, True), (81, 816, 136, True), (324, 288, 48, False), (1296, 144, 32, False),
                                             (49, 40, 24, False), (10, 24, 24, True), (25, 1392, 232, True), (7, 8, 100, False)])
def test_conv1x1_gated_bn_vs_torch(dev, ops, hw, cin, cout, res):
    """sigmoid(x_squeezed) * x -> _project_conv -> _bn2 (+ identity): the gate is applied to the A operand inside the GEMM."""
    n = 3
    side = int(round(hw ** 0.5))
    hh, ww = (side, side) if side * side == hw else (1, hw)
    x = rnd((n, hh, ww, cin), 720 + cin)
    gate = torch.sigmoid(rnd((n, cin), 721))
    w = rnd((cout, cin), 722, (1.0 / cin) ** 0.5)
    scale, bias = rnd((cout,), 723, 0.2) + 1.0, rnd((cout,), 724, 0.1)
    r = rnd((n, hh, ww, cout), 725) if res else None
    ref = ((x * gate.view(n, 1, 1, cin)).reshape(-1, cin).double() @ w.t().double()).float().view(n, hh, ww, cout) * scale + bias
    if res:
        ref = ref + r
    got = ops.conv1x1_gated_bn(x.to(dev), gate.to(dev), w.to(dev), scale.to(dev), bias.to(dev), r.to(dev) if res else None)
    assert (got.cpu() - ref).abs().max().item() < 2e-5 * max(1.0, float(ref.abs().max()))
    # gate = None: a plain 1x1 conv + BN
    plain = ops.conv1x1_gated_bn(x.to(dev), None, w.to(dev), scale.to(dev), bias.to(dev))
    ref_p = (x.reshape(-1, cin).double() @ w.t().double()).float().view(n, hh, ww, cout) * scale + bias
    assert (plain.cpu() - ref_p).abs().max().item() < 2e-5 * max(1.0, float(ref_p.abs().max()))
    if cin % 8 == 0:
        x16, w16 = x.half(), w.half()
        xg = (x16.float() * gate.view(n, 1, 1, cin)).half()            # the gated operand is rounded to fp16 before the MFMA
        ref16 = (xg.reshape(-1, cin).double() @ w16.t().double()).float().view(n, hh, ww, cout) * scale + bias
        if res:
            ref16 = ref16 + r.half().float()
        got16 = ops.conv1x1_gated_bn(x16.to(dev), gate.to(dev), w16.to(dev), scale.to(dev), bias.to(dev), r.half().to(dev) if res else None)
        assert got16.dtype == torch.float16
        assert (got16.float().cpu() - ref16).abs().max().item() < 2e-3 * max(1.0, float(ref16.abs().max()))


def test_conv_engine_swish_epilogue(dev, ops):
    """ADAF_ACT_SWISH in the MFMA engine's epilogue (expand / head convs): x * sigmoid(x) after the BN affine."""
    x = rnd((2, 9, 9, 96), 730)
    w = rnd((576, 96, 1, 1), 731, 0.1)
    scale, bias = rnd((576,), 732, 0.2) + 1.0, rnd((576,), 733, 0.1)
    ref = _swish((x.reshape(-1, 96) @ w.view(576, 96).t()) * scale + bias).view(2, 9, 9, 576)
    wp = ops.pack_conv_weight(w.to(dev))
    got = ops.conv2d_bn_act(x.to(dev), wp, scale.to(dev), bias.to(dev), act=ops.ACT_SWISH)
    assert (got.cpu() - ref).abs().max().item() < 2e-5
    naive = ops.conv2d_bn_act(x.to(dev), wp, scale.to(dev), bias.to(dev), act=ops.ACT_SWISH, naive=True)
    assert (naive.cpu() - ref).abs().max().item() < 2e-5
    got16 = ops.conv2d_bn_act_f16(x.half().to(dev), ops.pack_conv_weight_f16(w.to(dev)), scale.to(dev), bias.to(dev), act=ops.ACT_SWISH)
    ref16 = _swish((x.half().float().reshape(-1, 96) @ w.half().float().view(576, 96).t()) * scale + bias).view(2, 9, 9, 576)
    assert (got16.float().cpu() - ref16).abs().max().item() < 2e-3 * float(ref16.abs().max())


# ------------------------------------------------------------------------------------ the network
def _net(dev, name, classes, dtype="f32", seed=1007, image_size="native"):
    from adafocus_amd.efficientnet import EfficientNet
    m = EfficientNet.from_name(name, num_classes=classes, image_size=image_size, dtype=dtype).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed).items()}
    m.load_state_dict(sd, strict=True)
    return m.to(dev), sd


@pytest.mark.parametrize("size,image_size", [(144, "native"), (96, "native"), (100, "native"), (100, None), (75, None)])
def test_b3_fp32_storage_vs_oracle(dev, R, size, image_size):
    """EfficientNet-B3, fp32 storage: every block boundary and the final features against the CPU restatement, at BASELINE
    config 5's 144^2 patches and at 96^2 / 100^2 with the package's default padding (static, computed for the native 300^2:
    100^2 then runs on 50/25/13/6/3 maps), and with the package's dynamic padding at sizes whose maps go odd (100: 50/25/13/7/4;
    75: 38/19/10/5/3) where the SAME padding turns asymmetric in different places."""
    m, sd = _net(dev, "efficientnet-b3", 200, image_size=image_size)
    x = _smooth((2, 3, size, size), 740 + size)
    from adafocus_amd.utils import nchw_to_nhwc4
    x4 = nchw_to_nhwc4(x.to(dev))
    eng = m.engine()
    with torch.no_grad():
        for upto in (0, 1, 2, 3, 5, 6, 8, 9, 13, 14, 18, 19, 24, 26):
            ref = R.extract_features(sd, x, "efficientnet-b3", image_size=image_size, upto=upto)
            got = eng.forward_blocks(x4, upto).cpu().permute(0, 3, 1, 2)
            assert got.shape == ref.shape, (upto, got.shape, ref.shape)
            assert _close(got, ref)[0], (size, upto, _close(got, ref)[1])
        ref_map = R.extract_features(sd, x, "efficientnet-b3", image_size=image_size)
        ref_vec = R.features_pooled(sd, x, "efficientnet-b3", image_size=image_size)
        fmap = m.extract_features(x.to(dev)).cpu()
        fvec = m.features_nhwc4(x4).cpu()
        pooled = m.get_featmap(x.to(dev), pooled=True).cpu()
    assert fmap.shape == ref_map.shape and _close(fmap, ref_map)[0]
    assert fvec.shape == (2, 1536) and _close(fvec, ref_vec)[0]
    assert torch.equal(pooled.view(2, -1), fvec)
    assert (ref_vec[0] - ref_vec[1]).abs().max().item() > 5 * TOL           # the two samples really differ


def test_b3_logits_and_static_padding_for_another_resolution(dev, R):
    """forward() = pooled features -> _fc; and EfficientNet.from_name(..., image_size=300) run on 144^2 input: the SAME
    padding is the one computed for the 300^2 chain (75 -> 38 pads the 5x5 / stride-2 conv (2, 2), the 36-pixel map would
    get (1, 2))."""
    m, sd = _net(dev, "efficientnet-b3", 200)
    x = _smooth((2, 3, 144, 144), 750)
    with torch.no_grad():
        ref = R.features_pooled(sd, x, "efficientnet-b3") @ sd["_fc.weight"].t() + sd["_fc.bias"]
        got = m(x.to(dev)).cpu()
    assert _close(got, ref)[0]
    # the default IS the package's: static padding for the native 300^2; image_size=None = the package's dynamic padding
    mdyn, sddyn = _net(dev, "efficientnet-b3", 200, image_size=None)
    assert m.image_size == 300 and mdyn.image_size is None
    with torch.no_grad():
        ref300 = R.extract_features(sddyn, x, "efficientnet-b3", image_size=300)
        refdyn = R.extract_features(sddyn, x, "efficientnet-b3", image_size=None)
        gotdyn = mdyn.extract_features(x.to(dev)).cpu()
        got300 = m.extract_features(x.to(dev)).cpu()
    assert got300.shape == ref300.shape and _close(got300, ref300)[0]
    assert gotdyn.shape == refdyn.shape and _close(gotdyn, refdyn)[0]
    assert (ref300 - refdyn).abs().max().item() > 10 * TOL


def test_b0_fp32_storage_vs_oracle(dev, R):
    """The same object builds every member of the family from its (width, depth) coefficients."""
    m, sd = _net(dev, "efficientnet-b0", 10, seed=5)
    x = _smooth((3, 3, 128, 128), 760)
    with torch.no_grad():
        ref = R.extract_features(sd, x, "efficientnet-b0")
        got = m.extract_features(x.to(dev)).cpu()
    assert got.shape == ref.shape == (3, 1280, 4, 4) and _close(got, ref)[0]


def test_b3_fp16_storage_vs_oracle(dev, R):
    """fp16 STORAGE of activations and 1x1 filters (fp32 accumulate / BN / swish / SE): measured error against the fp32
    oracle on the 144^2 patches of config 5.  fp16 carries 11 significant bits: each stored map is rounded by 2^-11
    relative, and 80 layers of that accumulate to ~1e-2 relative rms at the far end (the network itself, in fp32 storage,
    lands 1e-6 from the oracle -- test above)."""
    m16, sd = _net(dev, "efficientnet-b3", 200, dtype="f16")
    m32, _ = _net(dev, "efficientnet-b3", 200, dtype="f32")
    x = _smooth((4, 3, 144, 144), 770)
    from adafocus_amd.utils import nchw_to_nhwc4
    x4 = nchw_to_nhwc4(x.to(dev))
    with torch.no_grad():
        ref = R.features_pooled(sd, x, "efficientnet-b3")
        v16 = m16.features_nhwc4(x4).cpu()
        v32 = m32.features_nhwc4(x4).cpu()
        b16 = m16.engine().forward_blocks(x4, 5)
        refb = R.extract_features(sd, x, "efficientnet-b3", upto=5)
    assert b16.dtype == torch.float16
    relb = ((b16.float().cpu().permute(0, 3, 1, 2) - refb).pow(2).mean().sqrt() / refb.pow(2).mean().sqrt()).item()
    rel = ((v16 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print("effnet-b3 fp16 storage: rel rms after 5 blocks %.2e, pooled features %.2e, max abs %.2e" % (relb, rel, (v16 - ref).abs().max().item()))
    assert not torch.equal(v16, v32)                        # the fp16 plan really ran
    assert _close(v32, ref)[0]
    assert relb < 5e-3 and rel < 3e-2
    assert (v16 - ref).abs().max().item() < 5e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("size,image_size,whole", [(144, "native", 16), (100, "native", 15), (75, "native", 11), (100, None, 15), (75, None, 15),
                                                   (128, None, 15)])
def test_b3_whole_block_kernels_equal_the_four_launch_plan(dev, size, image_size, whole):
    """fp16 storage: the stride-1 MBConv blocks on maps up to 9 x 9 as ONE launch each (csrc/mbconv_whole.hip: expand ->
    depthwise out of the accumulators -> in-block squeeze-and-excite -> gated project) against the four-launch plan, at every
    block boundary behind a fused block and at the pooled features.  Every stored value comes from the same arithmetic with the
    same fp16 roundings; the squeeze adds its pixels in another order, which can move a gated value across an fp16 rounding
    boundary: agreement to a few fp16 ulps.  144^2 (config 5): blocks 9-17 on 9 x 9, block 18 (stride 2: 9 x 9 -> 5 x 5) and 19-24 on 5 x 5 maps (block 25, hid 2304, keeps the
    four-launch plan); the other sizes /
    padding rules put 3 x 3 ... 8 x 8 maps under the kernel (native padding at 100^2: 6 x 6 and 3 x 3; at 75^2: 9 x 9 for blocks
    6-7 and 4 x 4; dynamic padding at 100^2: 7 x 7 and 4 x 4; at 75^2: 5 x 5 and 3 x 3; at 128^2: 8 x 8 and 4 x 4); n = 5 leaves a
    ragged last image pair where a workgroup owns two images."""
    from adafocus_amd.utils import nchw_to_nhwc4
    x4 = nchw_to_nhwc4(_smooth((5, 3, size, size), 800 + size).to(dev))
    m, _ = _net(dev, "efficientnet-b3", 200, dtype="f16", image_size=image_size)
    cuts = (7, 8, 9, 10, 13, 14, 15, 18, 19, 20, 24, 25, 26)
    with torch.no_grad():
        m.fusion = True
        assert m.engine().whole_blocks(size) == whole
        fused = [m.engine().forward_blocks(x4, k).float().clone() for k in cuts] + [m.features_nhwc4(x4).clone()]
        m.fusion = False
        assert m.engine().whole_blocks(size) == 0
        plain = [m.engine().forward_blocks(x4, k).float().clone() for k in cuts] + [m.features_nhwc4(x4).clone()]
    for k, f, p in zip(cuts + ("features",), fused, plain):
        assert f.shape == p.shape
        assert (f - p).abs().max().item() <= 4e-3 * max(1.0, float(p.abs().max())), (size, k, (f - p).abs().max().item())
    m32, _ = _net(dev, "efficientnet-b3", 200, dtype="f32")
    assert m32.engine().whole_blocks(size) == 0          # fp32 storage keeps the four-launch plan


@pytest.mark.parametrize("size,image_size", [(144, "native"), (100, "native"), (75, "native"), (100, None), (75, None), (128, None)])
def test_b3_fused_expand_depthwise_equals_the_two_launch_plan(dev, size, image_size):
    """fp16 storage: the expand conv computed inside the depthwise launch's staging step (dw_same_kernel XN > 0: blocks 2-8, no expanded
    map in HBM) against expand launch + depthwise launch, at every block boundary behind a fused block, on a smooth and on a noise batch.
    Same arithmetic per stored value (the 16x16x32 MFMA here and the strip kernel's two 32x32x16 give the same bits: where the two tile
    plans coincide -- 75^2 with dynamic padding -- the networks agree exactly); the fused launch's tile plan, hence the order of the squeeze
    partial sums, is its own, which can move a gated value across an fp16 rounding: agreement to a few fp16 ulps.  The sizes put odd
    maps (25, 13, 7), partial tiles, asymmetric SAME padding and both k-step counts (cin 24 / 32 / 48) under the kernel; n = 5 leaves
    ragged image groups."""
    from adafocus_amd import _lib as L
    from adafocus_amd.utils import nchw_to_nhwc4
    m, _ = _net(dev, "efficientnet-b3", 200, dtype="f16", image_size=image_size)
    g = torch.Generator().manual_seed(5100 + size)
    for x in (_smooth((5, 3, size, size), 900 + size), torch.randn((5, 3, size, size), generator=g) * 0.5):
        x4 = nchw_to_nhwc4(x.to(dev))
        with torch.no_grad():
            eng = m.engine()
            assert int(L.get_option("effnet_plan")) & L.EF_PLAN_FUSED_EXPAND 
            # blocks 2-8, less those small enough for the whole-image kernel (75^2 with the native padding: blocks 6-7 sit on 9 x 9 maps)
            assert eng.fused_expand_blocks(size) == (5 if (size, image_size) == (75, "native") else 7)
            fused = [eng.forward_blocks(x4, k).float().clone() for k in range(3, 10)] + [m.features_nhwc4(x4).clone()]
            with L.option("effnet_plan", int(L.get_option("effnet_plan")) - L.EF_PLAN_FUSED_EXPAND):
                assert eng.fused_expand_blocks(size) == 0
                plain = [eng.forward_blocks(x4, k).float().clone() for k in range(3, 10)] + [m.features_nhwc4(x4).clone()]
        for k, (f, p) in enumerate(zip(fused, plain)):
            assert f.shape == p.shape
            assert (f - p).abs().max().item() <= 4e-3 * max(1.0, float(p.abs().max())), (size, k, (f - p).abs().max().item())


def test_b3_whole_block_kernel_single_block_vs_torch(dev, R):
    """One fused block against the oracle's mbconv on the block's own fp16 input (so only this block's arithmetic is compared):
    block 14 (5x5 window, 9 x 9 map, identity skip) and block 24 (3x3 window, 5 x 5 map, 232 -> 384, no skip), 3 images."""
    from adafocus_amd.utils import nchw_to_nhwc4
    x4 = nchw_to_nhwc4(_smooth((3, 3, 144, 144), 860).to(dev))
    m, sd = _net(dev, "efficientnet-b3", 200, dtype="f16")
    with torch.no_grad():
        eng = m.engine()
        for bi in (14, 24):
            xin = eng.forward_blocks(x4, bi)                       # fp16 NHWC input of block bi
            got = eng.forward_blocks(x4, bi + 1).float().cpu().permute(0, 3, 1, 2)
            ref = R.mbconv_block(sd, xin.float().cpu().permute(0, 3, 1, 2), "efficientnet-b3", bi)
            assert got.shape == ref.shape
            rel = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            assert rel < 2e-3, (bi, rel)                          # fp16 storage of E, D and the output: ~2^-11 each


def test_b3_batch_invariance_and_chunking(dev, monkeypatch):
    """A frame's features do not depend on what else is in the batch (deterministic squeeze sums, no atomics), in both
    storage modes; rows written into a strided `out` view match."""
    for dtype in ("f32", "f16"):
        m, _ = _net(dev, "efficientnet-b3", 200, dtype=dtype)
        x = _smooth((5, 3, 144, 144), 780)
        from adafocus_amd.utils import nchw_to_nhwc4
        x4 = nchw_to_nhwc4(x.to(dev))
        with torch.no_grad():
            full = m.features_nhwc4(x4).clone()
            again = m.features_nhwc4(x4).clone()
            one = m.features_nhwc4(x4[3:4]).clone()
            buf = torch.zeros((5, 1280 + 1536), device=dev)
            m.features_nhwc4(x4, out=buf[:, 1280:])
        assert torch.equal(full, again) and torch.equal(full[3:4], one)
        assert torch.equal(buf[:, 1280:], full) and float(buf[:, :1280].abs().max()) == 0.0


def test_b3_every_position_in_the_batch_gives_the_same_bits(dev):
    """The rare-event form of the test above (round 4: a frame's fp16 features depended on which MFMA row its pixels landed on --
    the compiler had folded SOME fp16 conversions of an unrolled epilogue into single-rounding v_fma_mixlo_f16 -- about six elements in
    41 k after block 3, invisible to one smooth image).  Noise frames, every block boundary of the multi-launch part and two of the
    whole-image part, every position of a batch of five against the frame alone, and a batch of five copies against itself."""
    from adafocus_amd.utils import nchw_to_nhwc4
    for dtype in ("f16", "f32"):
        m, _ = _net(dev, "efficientnet-b3", 200, dtype=dtype)
        g = torch.Generator().manual_seed(4100)
        for trial in range(2 if dtype == "f16" else 1):
            x4 = nchw_to_nhwc4((torch.randn((5, 3, 144, 144), generator=g) * 0.5).to(dev))
            with torch.no_grad():
                m.features_nhwc4(x4)
                net = m.engine()
                for upto in (3, 4, 5, 6, 9, 12, 20, 26):
                    full = net.forward_blocks(x4, upto).clone()
                    for i in range(5):
                        one = net.forward_blocks(x4[i:i + 1].contiguous(), upto)
                        assert torch.equal(full[i:i + 1], one), (dtype, trial, upto, i, int((full[i:i + 1] != one).sum()))
                    rep = net.forward_blocks(x4[[3, 3, 3, 3, 3]].contiguous(), upto)
                    assert all(torch.equal(rep[i], rep[0]) for i in range(1, 5)), (dtype, trial, upto)
                # ... and the kernel form that claims the same bits agrees on noise too: the narrow-project strip kernel against the launch
                # it replaces (the whole-image blocks and the tiny-map depthwise kernel add their squeeze sums in another order, the
                # expand / stem strip kernels replace the generic engine, whose k order is its own: not part of the claim)
                from adafocus_amd import _lib as L
                want = net.forward_blocks(x4, 26).clone()
                for plan in (int(L.get_option("effnet_plan")) - L.EF_PLAN_STRIP_PROJECT,):
                    with L.option("effnet_plan", plan):
                        got = net.forward_blocks(x4, 26)
                    assert torch.equal(got, want), (dtype, trial, plan, int((got != want).sum()))


def test_b3_full_size_batch_with_a_ragged_chunk_equals_small_batches(dev):
    """BASELINE config 5's 1024 patches of 144^2 plus a ragged tail (1030 frames: one full 1024-frame chunk and a 6-frame one, odd image groups
    in every kernel that owns two images): frames from the head, the chunk boundary and the tail give the bits they give in a batch of three."""
    from adafocus_amd.utils import nchw_to_nhwc4
    m, _ = _net(dev, "efficientnet-b3", 200, dtype="f16")
    g = torch.Generator().manual_seed(4300)
    base = nchw_to_nhwc4((torch.randn((6, 3, 144, 144), generator=g) * 0.5).to(dev))
    x4 = base[torch.arange(1030, device=dev) % 6].contiguous()
    with torch.no_grad():
        full = m.features_nhwc4(x4)
        small = m.features_nhwc4(base[[1, 5, 3]].contiguous())           # frames 1, 5, 3
    assert torch.isfinite(full).all()
    for row in (1, 5, 3, 1023, 1025, 1027, 1029):           # head, last frame of the full chunk, the ragged chunk; row % 6 in {1, 5, 3}
        want = {1: 0, 5: 1, 3: 2}[row % 6]
        assert torch.equal(full[row], small[want]), (row, float((full[row] - small[want]).abs().max()))


# ------------------------------------------------------------------------------------ BASELINE config 5 as named
def _act_args(**over):
    class A:
        pass
    a = A()
    a.__dict__.update(num_segments=16, num_classes=200, reward="random", dataset="actnet", input_size=224, batch_size=2,
                      patch_size=144, with_glancer=True, feature_map_channels=1280, glance_size=224, action_dim=49,
                      hidden_state_dim=1024, policy_conv=True, gpu=0, continuous=False, gamma=0.7, policy_lr=0.0003,
                      random_patch=False, dropout=0.5, consensus="gru", hidden_dim=1024)
    a.__dict__.update(over)
    return a


def test_config5_efficientnet_b3_t16_p144_end_to_end(dev, R):
    """BASELINE config 5 as named: EfficientNet-B3 local CNN, T = 16, 144^2 patches, fp16 storage, through GFV.hot_path
    (gather -> local CNN -> concat with the glancer vector -> GRU classifier).  Checker: the oracle's gather + the
    EfficientNet restatement + the oracle's GRU classifier in fp32 (parity unpinned, see the module docstring); the same
    model in fp32 storage meets the fp32 bar."""
    from adafocus_amd.gfv_net import GFV
    from oracle import ref_model as O
    b, t = 3, 16
    fr = torch.cat([_smooth((1, 3 * t, 224, 224), 790 + i) for i in range(b)])
    _, act = synth.synth_actions(b * t, 7, seed=52)
    gvec = rnd((b, t, 1280), 53, 0.5)
    out = {}
    for dtype in ("f16", "f32"):
        m = GFV(_act_args(local_arch="efficientnet-b3", local_dtype=dtype)).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()}
        m.load_state_dict(sd, strict=True)
        m = m.to(dev)
        assert m.focuser.feature_dim == 1536 and m.classifier.gru.weight_ih_l0.shape[1] == 1280 + 1536
        with torch.no_grad():
            lg, last, feat = m.hot_path(fr.view(b * t, 3, 224, 224).to(dev), gvec.to(dev), torch.from_numpy(act).to(dev), b, t)
        out[dtype] = (lg.cpu(), last.cpu(), feat[:, :, 1280:].cpu())
    with torch.no_grad():
        patches = O.get_patch(fr.view(b * t, 3, 224, 224), torch.from_numpy(act), 144)
        esd = {k[len("focuser.net."):]: v for k, v in sd.items() if k.startswith("focuser.net.")}
        local = R.features_pooled(esd, patches, "efficientnet-b3").view(b, t, -1)
        rl, rlast = O.recurrent_classifier(sd, "classifier.", torch.cat([gvec, local], dim=2))
    lg32, last32, f32 = out["f32"]
    ok, worst = _close(f32.reshape(b * t, -1), local.reshape(b * t, -1))
    assert ok, worst
    assert (lg32 - rl).abs().max().item() < TOL and (last32 - rlast).abs().max().item() < TOL
    lg16, last16, f16 = out["f16"]
    relf = ((f16 - local).pow(2).mean().sqrt() / local.pow(2).mean().sqrt()).item()
    rell = ((lg16 - rl).pow(2).mean().sqrt() / rl.pow(2).mean().sqrt()).item()
    print("config 5 (EfficientNet-B3, T=16, P=144): fp16 storage rel rms local features %.2e, logits %.2e, max |dlogit| %.2e"
          % (relf, rell, (lg16 - rl).abs().max().item()))
    assert relf < 2e-3 and rell < 2e-3        # measured 1.4e-4 .. 4e-4 (bench.py `also.config5...`): a 10x regression fails


@pytest.mark.parametrize("size,image_size", [(100, "native"), (75, None), (144, "native")])
def test_b3_fp16_forward_is_the_same_from_run_to_run(dev, size, image_size):
    """The same batch through the network repeatedly, at every block boundary from block 9 on (whole-block kernels on 3 x 3 ... 9 x 9
    maps): bit-identical every time.  A version of csrc/mbconv_whole.hip whose filter loads landed in registers that MFMAs issued just
    before them still read (or whose tap loads landed in accumulator registers the compiler knew to be dead) differed from run to run
    in a few hundred values per forward -- inside every tolerance of the other tests, on most runs (tools/exp/effnet_determinism.py)."""
    from adafocus_amd.utils import nchw_to_nhwc4
    m, _ = _net(dev, "efficientnet-b3", 200, dtype="f16", image_size=image_size)
    g = torch.Generator().manual_seed(4200 + size)
    x4 = nchw_to_nhwc4((torch.randn((5, 3, size, size), generator=g) * 0.5).to(dev))
    with torch.no_grad():
        for k in (10, 14, 18, 20, 22, 24, 25, 26):
            ref = m.engine().forward_blocks(x4, k).clone()
            for _ in range(5):
                assert torch.equal(m.engine().forward_blocks(x4, k), ref), (size, k)
        ref = m.features_nhwc4(x4).clone()
        for _ in range(5):
            assert torch.equal(m.features_nhwc4(x4), ref), size


@pytest.mark.parametrize("name,dtype,size", [("efficientnet-b3", "f32", 144), ("efficientnet-b3", "f32", 100), ("efficientnet-b3", "f16", 75),
                                              ("efficientnet-b0", "f32", 96), ("efficientnet-b3", "f16", 144)])
def test_packed_stem_bit_identical_to_the_k36_stem(dev, name, dtype, size):
    """The stem with k packed to 27 (+ 1) and columns 32-47 on a 16-column MFMA tile (ef_stem_packed_kernel; B3's 40 channels: 14 x
    32x32x2 + 14 x 16x16x4 steps per band) against the K = 36 x 64-column form (option bit ADAF_EF_PLAN_PACKED_STEM off).  An fp32 MFMA
    adds its products in k order as a chain of fused multiply-adds (tools/exp/mfma_order_test.hip), the packed form visits the
    non-zero products in the same order and the dropped ones were exact zeros: torch.equal, in both storage modes, on ragged tiles
    (100, 75) and for a 32-channel stem (B0: no second tile)."""
    from adafocus_amd import _lib as L
    from adafocus_amd.utils import nchw_to_nhwc4
    m, _ = _net(dev, name, 200, dtype=dtype)
    x4 = nchw_to_nhwc4(_smooth((3, 3, size, size), 880 + size).to(dev))
    plan = int(L.get_option("effnet_plan"))
    assert plan & L.EF_PLAN_PACKED_STEM
    with torch.no_grad():
        new = m.engine().forward_blocks(x4, 0).float().clone()
        with L.option("effnet_plan", plan & ~L.EF_PLAN_PACKED_STEM):
            old = m.engine().forward_blocks(x4, 0).float().clone()
    assert new.shape == old.shape and torch.isfinite(new).all() and float(new.abs().max()) > 0.1
    assert torch.equal(new, old)


@pytest.mark.parametrize("name,size,n", [("efficientnet-b3", 144, 7), ("efficientnet-b3", 96, 30), ("efficientnet-b3", 128, 9),
                                         ("efficientnet-b0", 144, 3), ("efficientnet-b3", 224, 2)])
def test_head_conv_with_pool_in_its_epilogue_bit_identical(dev, name, size, n):
    """fp16 storage, pooled features: the head conv (1x1 + BN + swish) with the global average pool in its epilogue (conv_gemm.hip
    adaf_launch_conv_pool16, option bit ADAF_EF_PLAN_HEAD_POOL: whole images per 128-row tile, the activated tile parked in LDS, a
    thread per (image, 4 channels) adds the pixels in pixel order and divides) against head conv -> fp32 map -> avgpool_kernel: same
    MFMA instruction and k order, same epilogue arithmetic, same order of the pool's additions -- torch.equal.  25 / 9 / 16-pixel maps
    (5 / 14 / 8 images per tile: n leaves a ragged last tile, and at 96^2 more than two tiles); 7 x 7 maps (224^2) fill a tile to 77 %
    and keep the two launches, as does the unpooled map."""
    from adafocus_amd import _lib as L
    from adafocus_amd.utils import nchw_to_nhwc4
    m, _ = _net(dev, name, 200, dtype="f16")
    x4 = nchw_to_nhwc4(_smooth((n, 3, size, size), 1300 + size).to(dev))
    plan = int(L.get_option("effnet_plan"))
    assert plan & L.EF_PLAN_HEAD_POOL
    with torch.no_grad():
        new = m.features_nhwc4(x4).clone()
        fmap = m.extract_features(x4[..., :3].permute(0, 3, 1, 2).contiguous()).float()
        with L.option("effnet_plan", plan & ~L.EF_PLAN_HEAD_POOL):
            old = m.features_nhwc4(x4).clone()
    assert new.shape == old.shape and new.shape[0] == n
    assert torch.isfinite(new).all() and float(new.abs().max()) > 1e-3
    assert torch.equal(new, old)
    pooled = fmap.mean(dim=(2, 3))
    assert (pooled - new).abs().max().item() < 1e-5 * max(1.0, float(pooled.abs().max()))
    assert (new[0] - new[1]).abs().max().item() > 1e-3            # the images really differ


@pytest.mark.parametrize("dtype,size,n", [("f16", 75, 513), ("f16", 96, 520), ("f32", 75, 513)])
def test_half_chunk_pairs_bit_identical_to_one_chunk(dev, dtype, size, n):
    """A batch of >= 512 patches that fits one chunk travels as a PAIR of half chunks on two streams (option bit ADAF_EF_PLAN_PAIR_CHUNKS;
    the second half on a library-owned stream forked from / joined to the caller's by events, with its own half of the workspace): block
    outputs, the map and the pooled features equal the single pass bit for bit -- a patch's arithmetic does not depend on the chunk it
    travels in -- on odd halves (257 + 256: a ragged image pair / group at the seam), several times over, in both storage modes."""
    from adafocus_amd import _lib as L
    from adafocus_amd.utils import nchw_to_nhwc4
    m, _ = _net(dev, "efficientnet-b3", 200, dtype=dtype)
    x4 = nchw_to_nhwc4(_smooth((n, 3, size, size), 1700 + size + n).to(dev))
    plan = int(L.get_option("effnet_plan"))
    assert plan & L.EF_PLAN_PAIR_CHUNKS
    with torch.no_grad():
        with L.option("effnet_plan", plan & ~L.EF_PLAN_PAIR_CHUNKS):
            ref = [m.engine().forward_blocks(x4, k).float().clone() for k in (3, 9, 25)] + [m.features_nhwc4(x4).clone()]
            ref_map = m.engine().forward(x4, m.image_size or 0, want_map=True, want_vec=True)
            ref_map = [t.clone() for t in ref_map]
        for _ in range(3):
            got = [m.engine().forward_blocks(x4, k).float() for k in (3, 9, 25)] + [m.features_nhwc4(x4)]
            for k, g, r in zip((3, 9, 25, "features"), got, ref):
                assert torch.isfinite(g).all() and torch.equal(g, r), (dtype, size, n, k)
            fm, fv = m.engine().forward(x4, m.image_size or 0, want_map=True, want_vec=True)
            assert torch.equal(fm, ref_map[0]) and torch.equal(fv, ref_map[1])


def test_chunk_pairs_inside_a_captured_hot_path(dev):
    """The paired half chunks fork to a library-owned stream and join back inside adaf_effnet_forward: captured into a HIP graph
    (GFV.capture_hot_path: the fork / join events become graph dependencies) the step replays bit-identically to the eager launch,
    on new inputs, several times -- B = 32, T = 16: 512 patches, the smallest batch that pairs."""
    from adafocus_amd import _lib as L
    from adafocus_amd.gfv_net import GFV
    assert int(L.get_option("effnet_plan")) & L.EF_PLAN_PAIR_CHUNKS
    b, t = 32, 16
    m = GFV(_act_args(local_arch="efficientnet-b3", local_dtype="f16", patch_size=96, batch_size=b)).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()}, strict=True)
    m = m.to(dev)
    g = m.capture_hot_path(b, t, exclusive=True, check_every=1)      # (persistent GRU scan in the graph: the same kernels as the eager step)
    for seed in (71, 72, 73):
        fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=seed)).to(dev).view(b * t, 3, 224, 224)
        _, act = synth.synth_actions(b * t, 7, seed=seed + 10)
        act = torch.from_numpy(act).to(dev)
        gv = rnd((b, t, 1280), seed + 20, 0.5).to(dev)
        with torch.no_grad():
            lg, last, _ = m.hot_path(fr, gv, act, b, t)
            lg, last = lg.clone(), last.clone()
            for _ in range(2):
                glg, glast = g(fr, gv, act)
                torch.cuda.synchronize()
                assert torch.isfinite(glg).all() and torch.equal(glg, lg) and torch.equal(glast, last), seed


def test_chunk_pairs_from_more_caller_streams_than_helpers(dev):
    """The library keeps one helper stream per caller stream, 16 at most; a 17th caller stream gets none and its two half chunks follow
    one another on its own stream: 20 caller streams, the same features bit for bit from every one of them."""
    from adafocus_amd.utils import nchw_to_nhwc4
    m, _ = _net(dev, "efficientnet-b3", 200, dtype="f16")
    x4 = nchw_to_nhwc4(_smooth((512, 3, 75, 75), 1900).to(dev))
    with torch.no_grad():
        ref = m.features_nhwc4(x4).clone()
        torch.cuda.synchronize()
        for i in range(20):
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                got = m.features_nhwc4(x4)
            s.synchronize()
            assert torch.equal(got, ref), i
