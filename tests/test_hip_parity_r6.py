"""Round 6: the strip-walking front of the MobileNetV2 glancer (csrc/mbstrip.hip; SURVEY.md §8 a10 / f2,
ACT/models/mobilenet.py:42-68,71-148) against the wave-private tile kernels it replaces and the three-launch plan."""
import pytest
import torch

from adafocus_amd import _lib as L
from tests.helpers import rnd, synth_sd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the HIP path needs the MI355X"
    return torch.device("cuda:0")


def _glancer(dev, seed=505):
    from adafocus_amd.mobilenet import mobilenet_v2
    net = mobilenet_v2().eval()
    sd = {k: v for k, v in synth_sd("ACT", seed, "glancer.net.", keep_prefix=False).items() if not k.startswith("classifier")}
    net.load_state_dict(sd, strict=False)
    return net.to(dev)


def _frames(dev, n, size, seed):
    x4 = torch.zeros((n, size, size, 4), device=dev)
    x4[..., :3] = rnd((n, size, size, 3), seed).to(dev)
    return x4


def test_strip_kernels_bit_identical_to_tiles_and_to_the_three_launch_plan(dev):
    """Strips need maps whose side is a multiple of 14 (stem output; 28 for the stride-2 blocks): 224^2 takes every strip kernel (stem + b1,
    b2 .. b6, expand -> depthwise of b8 .. b13 on 14^2 maps), 448^2 the same with several strips per map everywhere (28^2 maps for b8 .. b13:
    left and right edge strips), 56 / 84 / 140 / 168 some of them with edge strips only, partial last row segments (28 / 8, 42 / 8) and ragged
    strides, 200 and 120 none (the tile kernels).  Every plan must produce the same feature map, bit for bit: strips, tiles, tiles without the whole-block
    kernels, the unfused three launches."""
    net = _glancer(dev)
    for n, size in ((5, 224), (3, 56), (2, 84), (3, 140), (2, 168), (2, 448), (3, 200), (4, 120), (520, 56)):
        x4 = _frames(dev, n, size, 600 + size)
        outs = []
        with torch.no_grad():
            for strip, fusion in ((1, True), (0, True), (1, 9), (0, False)):
                with L.option("mb_strip", strip):
                    net._engine.fusion = fusion
                    fm, fv = net.features_from_nhwc4(x4)
                    outs.append((fm.clone(), fv.clone()))
        net._engine.fusion = True
        for fm, fv in outs[1:]:
            assert torch.equal(fm, outs[0][0]) and torch.equal(fv, outs[0][1]), (n, size)
        assert outs[0][0].abs().max().item() > 0.1


def test_strip_kernels_run_to_run_and_batch_position(dev):
    """The strip kernels request their operands behind VALU reads of the MFMA chains that read the landing registers (a returning load is not
    interlocked against an in-flight MFMA): the same batch ten times must give the same bits, and a frame's features must not depend on its
    position in the batch (1024 frames = two chunks side by side on two streams; 5 frames = a partial block of strips)."""
    net = _glancer(dev, 506)
    x4 = _frames(dev, 1024, 224, 77)
    with torch.no_grad():
        ref = [t.clone() for t in net.features_from_nhwc4(x4)]
        for _ in range(10):
            fm, fv = net.features_from_nhwc4(x4)
            assert torch.equal(fm, ref[0]) and torch.equal(fv, ref[1])
        sub = torch.cat([x4[1000:1003], x4[5:7]])
        fm, fv = net.features_from_nhwc4(sub)
        assert torch.equal(fm[:3], ref[0][1000:1003]) and torch.equal(fm[3:], ref[0][5:7]) and torch.equal(fv[3:], ref[1][5:7])


def test_sth_glancer_with_temporal_shift_on_strips(dev):
    """The Something-Something glancer (temporal shift in front of the residual blocks' expand convs, STH/models/gfv_net.py:235-246) at 224^2:
    the expand -> depthwise strips of the 64- / 96-channel blocks take the shift inside their pixel loads (a buffer descriptor over the clip: a
    neighbour frame outside it reads as zeros), the whole-block strips read a shifted copy, the identity rows stay unshifted -- strips == tiles ==
    unfused."""
    from adafocus_amd.gfv_net_sth import Glancer
    from tests.test_state_dict_compat import sth_args
    gl = Glancer(sth_args()).eval()
    gl.load_state_dict(synth_sd("STH", 78, "glancer.", keep_prefix=False), strict=True)
    gl = gl.to(dev)
    x = rnd((16, 3, 224, 224), 56).to(dev)
    outs = []
    with torch.no_grad():
        for strip, fusion in ((1, True), (0, True), (0, False)):
            with L.option("mb_strip", strip):
                gl.net._engine.fusion = fusion
                fm, logit = gl(x)
                outs.append((fm.clone(), logit.clone()))
    gl.net._engine.fusion = True
    for fm, logit in outs[1:]:
        assert torch.equal(fm, outs[0][0]) and torch.equal(logit, outs[0][1])


@pytest.mark.parametrize("n,p", [(8, 96), (160, 96), (130, 128), (129, 144)])
def test_split_tiles_lean_k_loop_bit_identical(dev, n, p):
    """The split tiles' K loop with the pre-split planes addressed from a scalar base (conv_gemm.hip launch_glds, option "split_lean") against
    the pointer-per-lane form: the same DMA into the same LDS image, so the features must be equal bit for bit -- ragged row tiles (8 patches),
    position-major tiles with a partial image group (160 = 128 + 32, 130 = 128 + 2) and the strided downsample convs included.
    ACT/models/resnet.py:94-114,170-192."""
    from adafocus_amd.resnet import resnet50
    net = resnet50(num_classes=200).eval()
    net.load_state_dict(synth_sd("ACT", 1007, "focuser.net.", keep_prefix=False), strict=True)
    net = net.to(dev)
    net.set_math("split_bf16")
    x = rnd((n, 3, p, p), 900 + n).to(dev)
    outs = []
    with torch.no_grad():
        for lean in (1, 0, 1):
            with L.option("split_lean", lean):
                outs.append(net.get_featvec(x).clone())
        with L.option("split_lean", 1), L.option("split_stage1_f32", 0):      # every conv on the split tiles (64-channel stage 1 too)
            all_lean = net.get_featvec(x).clone()
        with L.option("split_lean", 0), L.option("split_stage1_f32", 0):
            all_ptr = net.get_featvec(x).clone()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(all_lean, all_ptr)


@pytest.mark.parametrize("n,p,seg", [(16, 96, 8), (24, 144, 12), (160, 128, 8), (21, 128, 3)])
def test_shifted_conv1_lean_k_loop_bit_identical(dev, n, p, seg):
    """A conv1 with the fused temporal shift on the lean K loop (conv_gemm.hip LEAN + SPECIAL, option "tsm_lean"): the activations arrive through the
    range-checked buffer form of the LDS DMA, which writes zeros for the rows at clip ends, instead of a per-lane source select -- the same bytes in
    the same LDS image, so the trunk's features must be equal bit for bit to the select form's, for whole clips, ragged row tiles and clips of 3
    frames (every frame a clip end for one of the two shifted folds).  STH/ops/temporal_shift.py:28-46, STH/models/tsn.py:215-241."""
    from adafocus_amd.resnet import resnet50
    net = resnet50(num_classes=200).eval()
    net.load_state_dict(synth_sd("ACT", 1007, "focuser.net.", keep_prefix=False), strict=True)
    net = net.to(dev)
    net.tsm_segments = seg
    x = rnd((n, 3, p, p), 700 + n).to(dev)
    outs = []
    with torch.no_grad():
        for lean in (1, 0, 1):
            with L.option("tsm_lean", lean):
                outs.append(net.get_featvec(x).clone())
        net.tsm_segments = 0
        plain = net.get_featvec(x).clone()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert not torch.equal(outs[0], plain)          # (the shift is really there)
