"""GPU parity tests added in round 5 (through the C ABI): the reference's SHIPPED Something-Something evaluation configuration
-- STH/evaluate.sh:5-15, STH/conf/evaluate.yaml:29-30: num_segments_glancer = 8, num_segments_focuser = 12, patch_size = 144 --
against G13 (tools/gen_golden_r2.py:gen_sth_shipped, the real reference's outputs).  Tg != Tf: the policy's state is 1280 * 8
channels, the gather emits 12 patches per clip from one (y, x), and the local CNN's fused temporal shift runs over clips of
TWELVE segments (not a power of two: STH/ops/temporal_shift.py:28-46 with n_segment = 12)."""
import hashlib

import numpy as np
import pytest
import torch

from adafocus_amd import synth
from tests.helpers import golden, synth_sd

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("trunk_math")]

TOL = 1e-3
TG, TF, P = 8, 12, 144


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def O():
    from oracle import ref_model
    return ref_model


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def _shipped_model(dev):
    from adafocus_amd.gfv_net_sth import GFV
    from tests.test_state_dict_compat import sth_args
    a = sth_args()
    a.gpu, a.num_segments_focuser, a.patch_size = 0, TF, P
    m = GFV(a).eval()
    m.focuser.net.base_model = torch.nn.Sequential(*list(m.focuser.net.base_model.children())[:-1])  # evaluate.py:83
    m.load_state_dict(synth_sd("STH", 1007), strict=True)
    pol = {k[len("policy."):]: v for k, v in synth_sd("STH_POLICY", 1007).items()}
    m.focuser.policy.policy_old.load_state_dict(pol)
    m.focuser.policy.policy.load_state_dict(pol)
    m.focuser.policy.policy_old.eval()
    m.focuser.policy.policy.eval()
    return m.to(dev), a


def _clips():
    gl = torch.from_numpy(synth.synth_frames(2, TG, 224, seed=3))
    fo = torch.from_numpy(synth.synth_frames(2, TF, 224, seed=13))
    return gl, fo


def test_sth_shipped_configuration_golden(dev):
    """glance + action_stage2 (policy-driven, with the reward baseline on the reference's recorded torch.rand draw; forced action) +
    action_stage3 at Tg = 8 / Tf = 12 / P = 144: logits within 1e-3 of the reference's, patches bit-exact (sha256 of all
    2 x 12 x 3 x 144 x 144 values), the policy's GRU state within 1e-3.  No escape hatch: the reference's crop origins sit
    >= 0.02 px from a pixel boundary (stored with the fixture)."""
    g = golden("g13_sth_shipped")
    m, a = _shipped_model(dev)
    assert m.focuser.net.num_segments == TF and m.glancer.net.tsm_segments == TG
    gl, fo = _clips()
    gl, fo = gl.to(dev), fo.view(2, TF, 3, 224, 224).to(dev)
    forced = torch.from_numpy(g["forced_action"]).to(dev)
    with torch.no_grad():
        fm, glog = m.glance(gl)
        assert fm.shape == (2, TG, 1280, 7, 7) and glog.shape == (2, TG, 174)
        assert np.abs(glog.cpu().numpy() - g["glancer_logit"]).max() < TOL
        pred, base, patch = m.action_stage2(fo, fm, glog, 0, a, prev_local_patch=None, training=False,
                                            baseline_action=torch.from_numpy(g["rand"]).to(dev))
        hid = m.focuser.memory.hidden[-1]
        act = m.focuser.policy.policy_old.act_nhwc(fm.permute(0, 1, 3, 4, 2).reshape(2 * TG, 7, 7, 1280), 2, TG)
        pred3, patch3 = m.action_stage3(fo, fm, glog, 0, a, prev_local_patch=None)
        pred_f, base_f, patch_f = m.action_stage2(fo, fm, glog, 0, a, prev_local_patch=None, training=False, forced_action=forced,
                                                  baseline_action=torch.from_numpy(g["rand_forced"]).to(dev))
        pred3_f, patch3_f = m.action_stage3(fo, fm, glog, 0, a, prev_local_patch=None, forced_action=forced)
        nb, none, patch_nb = m.action_stage2(fo, fm, glog, 0, a, prev_local_patch=None, training=False, with_baseline=False)
    assert patch.shape == (2, TF, 3, P, P) and base.shape == (2, 174)
    assert g["policy_action_px_margin"].min() >= 0.02
    assert np.abs(act.cpu().numpy() - g["policy_action"]).max() < 1e-4
    assert np.abs(hid[0].cpu().numpy() - g["hidden"]).max() < TOL
    assert np.array_equal(_sha(patch.cpu().numpy()), g["patch_sha"])
    assert np.array_equal(patch[:, :, :, :4, :4].cpu().numpy(), g["patch_corner"])
    assert np.abs(pred.cpu().numpy() - g["logits"]).max() < TOL
    assert np.abs(base.cpu().numpy() - g["baseline"]).max() < TOL
    assert np.abs(pred3.cpu().numpy() - g["logits_stage3"]).max() < TOL and torch.equal(patch3, patch)
    assert np.array_equal(_sha(patch_f.cpu().numpy()), g["patch_forced_sha"])
    assert np.abs(pred_f.cpu().numpy() - g["logits_forced"]).max() < TOL
    assert np.abs(base_f.cpu().numpy() - g["baseline_forced"]).max() < TOL
    assert np.abs(pred3_f.cpu().numpy() - g["logits_stage3_forced"]).max() < TOL and torch.equal(patch3_f, patch_f)
    # the baseline branch is a second half of the same trunk pass: clips of 12 must not shift into the other half
    assert none is None and torch.equal(patch_nb, patch) and torch.equal(nb, pred)


def test_sth_shipped_tsm_trunk_clip_independence(dev):
    """A 12-segment temporal shift touches only the frames of its own clip: the features of clip k in a 5-clip batch (60 patches of
    144^2: ragged against every tile) are those of clip k run alone -- bit for bit (fp32 MFMA chains in one k order)."""
    from adafocus_amd.tsn import TSN
    from tests.helpers import rnd
    m, _ = _shipped_model(dev)
    net = m.focuser.net
    assert isinstance(net, TSN)
    x = rnd((5 * TF, P, P, 4), 77).to(dev)
    x[..., 3] = 0
    with torch.no_grad():
        full = net.features_nhwc4(x)
        alone = [net.features_nhwc4(x[k * TF:(k + 1) * TF].contiguous()) for k in (0, 3, 4)]
    assert full.shape == (5 * TF, 2048)
    for k, f in zip((0, 3, 4), alone):
        assert torch.equal(full[k * TF:(k + 1) * TF], f), k


def test_sth_shipped_trunk_vs_oracle(dev, O):
    """The TSM-ResNet-50 of the shipped configuration (12 segments, 144^2) against the oracle's trunk on 2 clips, through TSN's
    reference-layout forward (STH/models/tsn.py:215-241, no_reshape=True)."""
    from tests.helpers import rnd
    m, _ = _shipped_model(dev)
    sd = synth_sd("STH", 1007)
    sd = O.canonical_resnet_keys(sd, "focuser.net.base_model.")
    x = rnd((2 * TF, 3, P, P), 78)
    with torch.no_grad():
        got = m.focuser.net(x.to(dev), no_reshape=True).cpu()
        ref = O.resnet50_trunk(sd, "focuser.net.base_model.", x, TF, 8).flatten(1)
    assert got.shape == ref.shape == (2 * TF, 2048)
    assert (got - ref).abs().max().item() < 3e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("with_baseline", [True, False])
def test_validate_sth_shipped_configuration(dev, O, with_baseline):
    """evaluate.validate_sth (the loop of STH/evaluate.py:165-226) at the shipped configuration: from the loader's fp32 clips the last
    step's logits are G13's; from stacked uint8 clips (two streams of DIFFERENT length: 8 x 3 and 12 x 3 channels) they are
    torch.equal to the same clips normalised on the host."""
    from adafocus_amd import evaluate as E
    g = golden("g13_sth_shipped")
    m, a = _shipped_model(dev)
    a.batch_size, a.glance_size = 2, 224
    gl, fo = _clips()
    labels = torch.tensor([5, 100])

    class DS:
        def __init__(self, g_, f_, n):
            self.g, self.f, self.n = g_, f_, n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return self.g[i], self.f[i], labels[i % 2]

    torch.manual_seed(5)
    r = E.validate_sth(DS(gl, fo, 2), m, torch.nn.CrossEntropyLoss(), a, quiet=True, with_baseline=with_baseline, return_logits=True)
    assert np.abs(r[4].numpy() - g["logits"]).max() < TOL
    assert np.abs(m.focuser.memory.hidden[-1][0].cpu().numpy() - g["hidden"]).max() < TOL
    assert torch.equal(r[5], labels) and len(r[2]) == 1
    assert (r[2][0] is None) != with_baseline and (not with_baseline or np.isfinite(r[2][0]))
    ref1 = float(E.accuracy(torch.from_numpy(g["logits"]), labels)[0])
    assert abs(r[0] - ref1) < 1e-4

    gen = np.random.Generator(np.random.PCG64([23, 5]))
    n = 3                                                                           # ragged last batch
    gu = gen.integers(0, 256, size=(n, 224, 224, 3 * TG), dtype=np.uint8)
    fu = gen.integers(0, 256, size=(n, 224, 224, 3 * TF), dtype=np.uint8)
    gf = torch.stack([O.ingest_uint8(gu[i]) for i in range(n)])
    ff = torch.stack([O.ingest_uint8(fu[i]) for i in range(n)])
    assert gf.shape == (n, 3 * TG, 224, 224) and ff.shape == (n, 3 * TF, 224, 224)
    torch.manual_seed(11)
    r8 = E.validate_sth(DS(torch.from_numpy(gu), torch.from_numpy(fu), n), m, torch.nn.CrossEntropyLoss(), a, quiet=True,
                        with_baseline=with_baseline, return_logits=True)
    torch.manual_seed(11)
    r32 = E.validate_sth(DS(gf, ff, n), m, torch.nn.CrossEntropyLoss(), a, quiet=True, with_baseline=with_baseline, return_logits=True)
    assert torch.equal(r8[4], r32[4]) and torch.equal(r8[5], r32[5]) and r8[:3] == r32[:3]
    # ... and against the oracle on the fp32 clips (main branch)
    sd = synth_sd("STH", 1007)
    sd.update(synth_sd("STH_POLICY", 1007))
    sd = O.canonical_resnet_keys(sd, "focuser.net.base_model.")
    with torch.no_grad():
        ref, _, _ = O.sth_forward(sd, gf[:2], ff[:2].view(2, TF, 3, 224, 224), P, TG, TF)
    assert (r32[4][:2] - ref).abs().max().item() < TOL * max(1.0, ref.abs().max().item())


def test_sth_shipped_glancer_and_focuser_in_one_model_forward(dev, O):
    """GFV.forward (STH/models/gfv_net.py:74-99, eval): the TSM glancer at 8 segments beside the TSM focuser at 12 in one model --
    the two shift lengths live in different sub-modules and must not leak into each other."""
    m, _ = _shipped_model(dev)
    g = golden("g13_sth_shipped")
    gl, fo = _clips()
    with torch.no_grad():
        out = m(input=fo.to(dev), scan=gl.to(dev))
    assert out.shape == (2, 174)
    assert np.abs(out.cpu().numpy() - g["logits"]).max() < TOL


def test_shift_place_block_golden_and_oracle(dev, O):
    """shift_place = 'block' (STH/ops/temporal_shift.py:104-121: TemporalShift around whole Bottlenecks -- conv1, the downsample conv and
    the identity read the shifted block input) on the HIP trunk (adaf_resnet50_set_shift_place): the reference's own features (G14),
    and against the oracle at 12 segments / 144^2 where every fused launch form is in play."""
    from adafocus_amd.tsn import TSN
    from tests.helpers import rnd
    g = golden("g14_sth_block_shift")
    net = TSN(4, base_model="resnet50", is_shift=True, shift_div=8, shift_place="block")
    net.base_model = torch.nn.Sequential(*list(net.base_model.children())[:-1])         # STH/evaluate.py:83
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1414).items()}
    net.load_state_dict(sd, strict=True)
    net = net.eval().to(dev)
    with torch.no_grad():
        got = net(rnd((8, 3, 64, 64), 141).to(dev), no_reshape=True).cpu()
    assert np.abs(got.numpy() - g["feat"]).max() < 3e-4 * max(1.0, float(np.abs(g["feat"]).max()))
    # 12 segments, 144^2, 24 patches; the same weights under 'blockres' give a different network
    csd = O.canonical_resnet_keys({"n." + k: v for k, v in sd.items()}, "n.base_model.")
    net.num_segments = net.base_model.tsm_segments = TF
    x = rnd((2 * TF, 3, P, P), 142)
    with torch.no_grad():
        got = net(x.to(dev), no_reshape=True).cpu()
        ref = O.resnet50_trunk(csd, "n.base_model.", x, TF, 8, shift_place="block").flatten(1)
        net.base_model.tsm_place = "blockres"
        res = net(x.to(dev), no_reshape=True).cpu()
        ref_res = O.resnet50_trunk(csd, "n.base_model.", x, TF, 8).flatten(1)
    scale = max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < 3e-4 * scale
    assert (res - ref_res).abs().max().item() < 3e-4 * scale
    assert (got - res).abs().max().item() > 1e-2


@pytest.mark.parametrize("b,t", [(33, 16), (64, 16), (65, 4), (200, 2)])
def test_gru_scan_two_slices_bit_identical_to_one(dev, b, t):
    """The persistent GRU scan cuts a batch of more than one m-tile into two slices, each scanned by its own 128 blocks with its own
    barrier words (option gru_scan_slices, the default since round 4).  Clips are independent: every logit and every hidden state is
    torch.equal to the single-slice scan -- with the folded classifier, with an initial state, and through the cooperative launch."""
    from adafocus_amd import _lib, hip_ops as ops
    from tests.helpers import rnd
    gen = {"gru.weight_ih_l0": (3072, 3328), "gru.weight_hh_l0": (3072, 1024), "gru.bias_ih_l0": (3072,), "gru.bias_hh_l0": (3072,),
           "fc.weight": (200, 1024), "fc.bias": (200,)}
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_state_dict(gen, 606).items()}
    args = (d["gru.weight_ih_l0"], d["gru.weight_hh_l0"], d["gru.bias_ih_l0"], d["gru.bias_hh_l0"], d["fc.weight"], d["fc.bias"])
    seq_args = (d["gru.weight_ih_l0"][:, :1024].contiguous(),) + args[1:4]
    x = rnd((b, t, 3328), 500 + b, 0.5).to(dev)
    xs = rnd((b, t, 1024), 600 + b, 0.5).to(dev)
    h0 = rnd((b, 1024), 700 + b, 0.5).to(dev)
    out = {}
    try:
        for mode in (1, 2):
            ops.set_gru_persistent(mode, dev)
            for slices in (1, 2):
                with _lib.option("gru_scan_slices", slices):
                    lg, last = [v.clone() for v in ops.gru_cls_forward(x, *args)]
                    hs = ops.gru_seq_forward(xs, *seq_args, h0=h0).clone()
                out[(mode, slices)] = (lg, last, hs)
    finally:
        ops.set_gru_persistent(1, dev)
    ref = out[(1, 1)]
    assert torch.isfinite(ref[0]).all() and torch.isfinite(ref[2]).all()
    for key, val in out.items():
        for a_, b_ in zip(ref, val):
            assert torch.equal(a_, b_), key
    assert ops.gru_scan_timeouts(dev) == 0


@pytest.mark.parametrize("p,n", [(96, 3), (96, 300), (64, 5), (128, 7), (144, 24), (128, 260)])
def test_stem_pool_strip_kernel_bit_identical_to_tile_kernel(dev, p, n):
    """csrc/stem.hip stem7x7_pool_rows_kernel (whole-width strips walked down the image, the shared conv row carried in registers, no halo
    recompute) against the tile kernel and against the unfused conv + max-pool launches: the same k order in every product, the same
    BN / ReLU / max -> torch.equal trunk features; image counts below, at and above one image per resident block."""
    from adafocus_amd import _lib
    from adafocus_amd.resnet import resnet50
    from tests.helpers import rnd
    net = resnet50(num_classes=200).eval()
    net.load_state_dict(synth_sd("ACT", 1007 + p, "focuser.net.", keep_prefix=False), strict=True)
    net = net.to(dev)
    net.set_math("f32")          # (the stem is on the fp32 pipe in either arithmetic; fusion on / off changes the split plan's stage 1)
    x = rnd((n, p, p, 4), 900 + p).to(dev)
    x[..., 3] = 0
    out = {}
    with torch.no_grad():
        for mode, fusion in ((2, 2), (0, 2), (0, 0)):
            with _lib.option("stem_rows", mode):
                net._sync().set_fusion(fusion)
                out[(mode, fusion)] = net.features_nhwc4(x).clone()
        net._sync().set_fusion(1)
    assert torch.isfinite(out[(2, 2)]).all() and out[(2, 2)].abs().max().item() > 0.1
    assert torch.equal(out[(2, 2)], out[(0, 2)]) and torch.equal(out[(2, 2)], out[(0, 0)])


@pytest.mark.parametrize("p,nf,fpa,sets,layout", [(96, 300, 1, 1, "nchw"), (96, 8, 1, 1, "nchw"), (96, 288, 12, 2, "nhwc4"), (128, 264, 8, 1, "nchw"),
                                                  (144, 264, 12, 2, "nchw"), (100, 260, 1, 1, "nhwc4"), (64, 520, 1, 1, "nhwc4")])
def test_trunk_from_frames_equals_gather_then_trunk(dev, p, nf, fpa, sets, layout):
    """adaf_resnet50_forward_frames -- get_patch (ACT/models/utils.py:37-51) folded into the trunk's first launch: the stem gathers its
    own windows from the planar or pixel-major frames at floor(action * (H - P)) -- gives torch.equal features to the gather launch
    followed by the trunk, with per-frame actions (ActivityNet), one action per clip (Something-Something), two action sets over the
    same frames (the reward baseline in one pass), and where the gathering stem does not apply (P = 100; fewer images than CUs)."""
    from adafocus_amd import hip_ops as ops
    from adafocus_amd.resnet import resnet50
    from adafocus_amd.utils import get_patch_nhwc4
    from tests.helpers import rnd
    net = resnet50(num_classes=200).eval()
    net.load_state_dict(synth_sd("ACT", 1007 + p, "focuser.net.", keep_prefix=False), strict=True)
    net = net.to(dev)
    net.tsm_segments = fpa if fpa > 1 else 0
    frames = rnd((nf, 3, 224, 224), 1300 + p).to(dev)
    gen = np.random.Generator(np.random.PCG64([p, nf]))
    act = gen.random((sets * nf // fpa, 2), dtype=np.float32)
    act[0], act[-1] = (0.0, 1.0), (1.0, 0.0)                    # the corners: origins 0 and H - P
    act = torch.from_numpy(act).to(dev)
    src = frames if layout == "nchw" else torch.cat([frames, torch.zeros_like(frames[:, :1])], 1).permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        got = net.features_from_frames(src, act, p, frames_per_action=fpa).clone()
        per = nf // fpa
        patches = torch.cat([get_patch_nhwc4(frames, act[g * per:(g + 1) * per], p, fpa) for g in range(sets)])
        ref = net.features_nhwc4(patches)
    assert got.shape == (sets * nf, 2048) and torch.isfinite(got).all()
    assert torch.equal(got, ref)
    if layout == "nhwc4":          # the same through the pixel-major gather
        p4 = torch.cat([ops.crop_gather_nhwc4(src, act[g * per:(g + 1) * per], p, fpa) for g in range(sets)])
        assert torch.equal(p4, patches)


@pytest.mark.parametrize("p,n", [(96, 256), (96, 140), (128, 130)])
def test_split_bf16_position_major_tap_skipping_bit_identical(dev, p, n):
    """Round 5: the split-bf16 tiles (pre-split weights) take position-major tiles with padding-tap skipping like the fp32 pipe's do.  A
    skipped tap is whole MFMA steps whose activation operand is zero in all three bf16 parts -- exact zeros into the fp32 accumulator --
    so the trunk's features must be torch.equal with the skipping on and off."""
    from adafocus_amd import hip_ops as ops
    from adafocus_amd.resnet import resnet50
    from tests.helpers import rnd
    net = resnet50(num_classes=200).eval()
    net.load_state_dict(synth_sd("ACT", 1007 + p, "focuser.net.", keep_prefix=False), strict=True)
    net = net.to(dev)
    net.set_math("split_bf16")
    x = rnd((n, p, p, 4), 1500 + p).to(dev)
    x[..., 3] = 0
    try:
        with torch.no_grad():
            ops.set_conv_pos_major(False, dev)
            ref = net.features_nhwc4(x).clone()
            ops.set_conv_pos_major(True, dev)
            got = net.features_nhwc4(x).clone()
    finally:
        ops.set_conv_pos_major(True, dev)
    assert torch.isfinite(got).all() and got.abs().max().item() > 0.1
    assert torch.equal(got, ref)


@pytest.mark.parametrize("p,n,tsm", [(96, 1024, 0), (96, 16, 0), (144, 96, 12), (128, 264, 8)])
def test_trunk_is_the_same_from_run_to_run(dev, p, n, tsm):
    """The same patches through the ResNet-50 trunk six times: bit-identical every time, in both arithmetic modes, at the headline batch
    (1024 patches: the batched tiles), at 16 patches (the small-batch conv form) and with the fused temporal shift.  (Why this is
    asked: EfficientNet's whole-block kernel once gave run-to-run differences inside every tolerance -- global loads that landed in
    registers of MFMAs still in flight, DESIGN 3.7.3.  The trunk's operands travel through LDS, not through such registers; this
    test keeps it that way.)"""
    from adafocus_amd.resnet import resnet50
    from tests.helpers import rnd
    net = resnet50(num_classes=200).eval()
    net.load_state_dict(synth_sd("ACT", 1007 + p, "focuser.net.", keep_prefix=False), strict=True)
    net = net.to(dev)
    net.tsm_segments = tsm
    x = rnd((n, 3, p, p), 5100 + p + n).to(dev)
    from adafocus_amd.utils import nchw_to_nhwc4
    x4 = nchw_to_nhwc4(x)
    with torch.no_grad():
        ref = net.features_nhwc4(x4).clone()
        for _ in range(5):
            assert torch.equal(net.features_nhwc4(x4), ref)


def test_glancer_is_the_same_from_run_to_run(dev):
    """The MobileNetV2 glancer (MFMA 1x1 convs fed straight from registers in its fused MBConv kernels) on the same 64 frames, six times."""
    from adafocus_amd.mobilenet import mobilenet_v2
    from tests.helpers import rnd
    net = mobilenet_v2().eval()
    sd = {k: v for k, v in synth_sd("ACT", 505, "glancer.net.", keep_prefix=False).items() if not k.startswith("classifier")}
    net.load_state_dict(sd, strict=False)
    net = net.to(dev)
    x = rnd((64, 3, 224, 224), 5200).to(dev)
    with torch.no_grad():
        fm, fv = [t.clone() for t in net.get_featmap(x)]
        for _ in range(5):
            fm2, fv2 = net.get_featmap(x)
            assert torch.equal(fm2, fm) and torch.equal(fv2, fv)


@pytest.mark.parametrize("n,tsm", [(520, 0), (512, 8), (520, 8), (600, 0)])
def test_glancer_half_chunk_pairs_equal_small_chunks(dev, n, tsm):
    """A glancer batch that fits one chunk but holds >= 512 frames travels as a PAIR of half chunks on the two streams (csrc/mobilenetv2.hip
    chunk_frames; whole clips per chunk under the temporal shift: 65 clips of 8 -> 33 + 32): the frames' map and pooled vector equal, bit
    for bit, what chunks of 128 frames give -- a frame's arithmetic does not depend on the chunk it travels in -- several times over (chunk limit
    1024 here so that odd halves occur; at the default limit of 512 the rule takes exactly 512 frames)."""
    from adafocus_amd import _lib as L
    from adafocus_amd.mobilenet import mobilenet_v2
    from adafocus_amd.utils import nchw_to_nhwc4
    from tests.helpers import rnd
    net = mobilenet_v2().eval()
    sd = {k: v for k, v in synth_sd("ACT", 505, "glancer.net.", keep_prefix=False).items() if not k.startswith("classifier")}
    net.load_state_dict(sd, strict=False)
    net = net.to(dev)
    x4 = nchw_to_nhwc4(rnd((n, 3, 64, 64), 5300 + n + tsm).to(dev))

    def run():
        out = net._engine.features(x4, tsm, 8) if tsm else net._engine.features(x4)
        return [t.clone() for t in out]
    with torch.no_grad():
        with L.option("mbv2_chunk", 128):
            ref = run()
        for _ in range(3):
            with L.option("mbv2_chunk", 1024):        # (the default, 512, pairs exactly 512 frames: the Something-Something batch)
                got = run()
            assert len(got) == len(ref) == 2
            for g, r in zip(got, ref):
                assert torch.isfinite(g).all() and float(g.abs().max()) > 1e-3 and torch.equal(g, r)


def test_glancer_from_more_caller_streams_than_helpers(dev):
    """adaf_mobilenetv2 keeps one second-chunk helper stream per caller stream, 16 at most; a 17th caller stream gets none (its chunks
    follow one another on its own stream) instead of an error: 20 caller streams, the same map and vector from every one of them."""
    from adafocus_amd.mobilenet import mobilenet_v2
    from adafocus_amd.utils import nchw_to_nhwc4
    from tests.helpers import rnd
    net = mobilenet_v2().eval()
    sd = {k: v for k, v in synth_sd("ACT", 505, "glancer.net.", keep_prefix=False).items() if not k.startswith("classifier")}
    net.load_state_dict(sd, strict=False)
    net = net.to(dev)
    x4 = nchw_to_nhwc4(rnd((512, 3, 64, 64), 5400).to(dev))
    with torch.no_grad():
        ref = [t.clone() for t in net._engine.features(x4)]
        torch.cuda.synchronize()
        for i in range(20):
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                got = net._engine.features(x4)
            s.synchronize()
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), i
