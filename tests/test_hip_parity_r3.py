"""GPU parity tests added in round 3 (through the C ABI): the small-batch ("latency") form of the conv engine -- csrc/conv_lat.hip,
v_mfma_f32_16x16x4_f32 chains that visit k in the engine's order -- must be BIT-IDENTICAL to the engine's kernels: per conv
(tile id 95 against the automatic choice and against the naive on-device kernel's tolerance), for the whole ResNet-50 trunk
(a patch's features do not depend on how many patches it travelled with: ACT/models/resnet.py:211-225 has no batch coupling),
and end to end for BASELINE config 1's step (B = 2, T = 8, P = 96) against the same clips inside a 64-clip batch."""
import pytest
import torch

from adafocus_amd import synth
from tests.helpers import rnd

pytestmark = [pytest.mark.gpu]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from adafocus_amd import hip_ops
    return hip_ops


CASES = [  # n, hw, cin, cout, k, stride, pad, residual, act
    (16, 6, 256, 256, 3, 1, 1, False, "relu"),       # stage 3 conv2
    (16, 12, 256, 256, 3, 2, 1, False, "relu"),      # stage 3.0 conv2 (stride 2)
    (16, 3, 512, 512, 3, 1, 1, False, "relu"),       # stage 4 conv2: 3x3 maps, most taps are padding
    (16, 3, 2048, 512, 1, 1, 0, False, "relu"),      # stage 4 conv1
    (16, 3, 512, 2048, 1, 1, 0, True, "relu"),       # stage 4 conv3 + identity
    (16, 6, 1024, 2048, 1, 2, 0, False, "none"),     # stage 4.0 downsample (1x1 / stride 2)
    (8, 24, 64, 64, 3, 1, 1, False, "relu"),         # stage 1 conv2 (cin = 64: 64-wide k slices)
    (3, 5, 128, 72, 3, 1, 1, True, "relu6"),         # ragged: 75 rows, 72 columns
    (1, 1, 64, 8, 1, 1, 0, False, "none"),           # one row
]


@pytest.mark.parametrize("n,hw,cin,cout,k,stride,pad,res,act", CASES)
def test_latency_form_conv_bit_identical_to_engine(dev, ops, n, hw, cin, cout, k, stride, pad, res, act):
    a = {"relu": ops.ACT_RELU, "relu6": ops.ACT_RELU6, "none": ops.ACT_NONE}[act]
    x = rnd((n, hw, hw, cin), 900 + hw + cin).to(dev)
    w = (rnd((cout, k, k, cin), 901 + cout) * 0.05).to(dev)
    sc, bi = (torch.rand(cout) + 0.5).to(dev), rnd((cout,), 902).to(dev)
    oh = (hw + 2 * pad - k) // stride + 1
    r = rnd((n, oh, oh, cout), 903).to(dev) if res else None
    eng = ops.conv2d_bn_act(x, w, sc, bi, r, stride, pad, a)
    lat = ops.conv2d_bn_act(x, w, sc, bi, r, stride, pad, a, tile=95)
    assert torch.equal(eng, lat)
    nai = ops.conv2d_bn_act(x, w, sc, bi, r, stride, pad, a, naive=True)
    assert float((lat - nai).abs().max()) <= 2e-4 * max(1.0, float(nai.abs().max()))


def test_latency_form_rejects_what_it_cannot_do(dev, ops):
    x = rnd((2, 8, 8, 24), 910).to(dev)               # cin % 64 != 0
    w = rnd((16, 1, 1, 24), 911).to(dev)
    with pytest.raises(RuntimeError):
        ops.conv2d_bn_act(x, w, None, None, None, 1, 0, ops.ACT_NONE, tile=95)


def _trunk(dev):
    from adafocus_amd.resnet import resnet50
    net = resnet50(num_classes=200).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
    return net.to(dev)


@pytest.mark.parametrize("patch", [96, 128])
def test_trunk_small_batches_equal_the_batched_plan(dev, patch):
    """1..48 patches take the latency form in stages 3 / 4 (GEMMs of <= 1536 rows), 64 patches the batched plan: same bits."""
    net = _trunk(dev)
    x = rnd((64, patch, patch, 4), 920 + patch).to(dev)
    x[..., 3] = 0
    t = net._sync()
    with torch.no_grad():
        big = t.forward(x).clone()
        for n in (1, 2, 8, 16, 33, 48):
            small = t.forward(x[:n].contiguous())
            assert torch.equal(small, big[:n]), n


def test_config1_step_equals_the_same_clips_in_a_large_batch(dev):
    """BASELINE config 1 (B = 2, T = 8, P = 96) through GFV.hot_path: logits of two clips alone == the logits of the same two
    clips as the first rows of a 16-clip batch (128 patches: every conv on the engine)."""
    from tests.test_hip_parity_r2 import _act_model
    m, _ = _act_model(dev)
    b, t = 16, 8
    fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=77)).to(dev).view(b * t, 3, 224, 224)
    _, act = synth.synth_actions(b * t, 7, seed=78)
    act = torch.from_numpy(act).to(dev)
    gv = rnd((b, t, 1280), 79, 0.5).to(dev)
    with torch.no_grad():
        lg_big, last_big, _ = m.hot_path(fr, gv, act, b, t)
        lg_big, last_big = lg_big.clone(), last_big.clone()
        lg, last, _ = m.hot_path(fr[:2 * t].contiguous(), gv[:2].contiguous(), act[:2 * t].contiguous(), 2, t)
    assert torch.equal(last, last_big[:2])
    assert torch.equal(lg.view(2, t, -1), lg_big.view(b, t, -1)[:2])


def test_average_pool_in_the_last_conv_epilogue_bit_identical(dev):
    """the trunk's last conv3 averages its map in its epilogue (whole images per row tile, pixel-order sum,
    division by hw: avgpool_kernel's arithmetic) -- same features bit for bit as conv3 + pooling launch, at 3x3 / 4x4 / 5x5 final
    maps, ragged image counts and with the temporal shift (tools/pool_ab.py under the library option "conv_pool" = 1 and 0)."""
    from adafocus_amd import _lib
    from tests.helpers import load_tool
    tool = load_tool("pool_ab")
    with _lib.option("conv_pool", 1):
        fused = tool.digests()
    with _lib.option("conv_pool", 0):
        plain = tool.digests()
    assert len(fused) == 5 and fused == plain
