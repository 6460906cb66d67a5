"""GPU parity tests added in round 2 (through the C ABI, against the oracle and the round-2 goldens of
tools/gen_golden_r2.py): resampling crop (N1), nearest glancer input, the per-step API surface (a3), BASELINE config 3's
shape end to end, Something-Something video_div = 2 + the reward-baseline branch (f4), the fused GRU scan (FC folded in,
h0, batches above 64, cooperative launch), the two-stream pipelined forward and multi-stream safety of the glancer."""
import hashlib
import os

import numpy as np
import pytest
import torch

from adafocus_amd import synth
from tests.helpers import golden, rnd, synth_sd

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("trunk_math")]

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
def O():
    from oracle import ref_model
    return ref_model


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def _to_nchw(x, layout, ops):
    if layout == ops.LAYOUT_NCHW:
        return x
    return x[..., :3].permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------ N1: crop-and-resize
def test_crop_resize_golden_all_layouts(dev, ops):
    g = golden("g10_resample")
    fr = rnd((4, 3, 224, 224), 101).to(dev)
    act = torch.from_numpy(g["actions"]).to(dev)
    for s_, p_ in g["cases"].tolist():
        for layout in (ops.LAYOUT_NCHW, ops.LAYOUT_NHWC, ops.LAYOUT_NHWC4):
            o = ops.crop_resize(fr, act, p_, size=s_, layout=layout)
            if layout == ops.LAYOUT_NHWC4:
                assert float(o[..., 3].abs().max()) == 0.0
            o = _to_nchw(o, layout, ops).cpu().numpy()
            assert np.abs(o[:, :, ::7, ::5] - g["sub_%d_%d" % (s_, p_)]).max() <= 1e-6, (s_, p_, layout)
            assert np.abs(o[:, :, -6:, -6:] - g["corner_%d_%d" % (s_, p_)]).max() <= 1e-6
            if s_ == p_:
                assert np.array_equal(_sha(o), g["sha_%d_%d" % (s_, p_)])
    sizes = torch.from_numpy(g["mixed_sizes"]).to(dev)
    o = ops.crop_resize(fr, act, 96, size=sizes).cpu().numpy()
    assert np.abs(o[:, :, ::7, ::5] - g["mixed_sub"]).max() <= 1e-6


def test_crop_resize_reduces_bit_exactly_to_the_gather(dev, ops):
    """size == patch: torch.equal with adaf_crop_gather_f32, through the launcher's short-cut AND through the resampling
    kernel itself (per-action sizes all equal to the patch size keep it on crop_resize_kernel)."""
    gen = np.random.Generator(np.random.PCG64([5, 77]))
    for p in (96, 128, 33):
        fr = torch.from_numpy(gen.standard_normal((6, 3, 224, 224), dtype=np.float32)).to(dev)
        fr[0, 0, 5, 7] = float("inf")          # zero-weight taps must not be touched (0 * inf would be NaN)
        fr[1, 2, 100, 100] = float("nan")
        act = torch.from_numpy(gen.random((6, 2), dtype=np.float32)).to(dev)
        act[0] = torch.tensor([0.0, 0.0])
        act[1] = torch.tensor([1.0, 1.0])
        for layout in (ops.LAYOUT_NCHW, ops.LAYOUT_NHWC4):
            ref, rc = ops.crop_gather(fr, act, p, 1, layout, return_coords=True)
            a, ca = ops.crop_resize(fr, act, p, size=None, layout=layout, return_coords=True)
            sizes = torch.full((6,), p, dtype=torch.int32, device=dev)
            b, cb = ops.crop_resize(fr, act, p, size=sizes, layout=layout, return_coords=True)
            for got in (a, b):
                assert torch.equal(torch.nan_to_num(got, 7.0, 8.0, 9.0), torch.nan_to_num(ref, 7.0, 8.0, 9.0))
            assert torch.equal(ca, rc) and torch.equal(cb, rc)
    # pixel-major frames in, pixel-major patches out
    fr4 = ops.crop_gather(fr, torch.zeros((6, 2), device=dev), 224, 1, ops.LAYOUT_NHWC4)
    ref = ops.crop_gather_nhwc4(fr4, act, 33)
    sizes = torch.full((6,), 33, dtype=torch.int32, device=dev)
    got = ops.crop_resize(fr4, act, 33, size=sizes, layout=ops.LAYOUT_NHWC4)
    assert torch.equal(torch.nan_to_num(got, 7.0, 8.0, 9.0), torch.nan_to_num(ref, 7.0, 8.0, 9.0))


def test_crop_resize_randomised_vs_oracle(dev, ops, O):
    gen = np.random.Generator(np.random.PCG64([6, 78]))
    for trial in range(24):
        hh = int(gen.choice([64, 96, 113, 224]))
        ww = hh + int(gen.choice([0, 0, 4, 7]))
        p = int(gen.integers(8, min(hh, 160) + 1))
        n, fpa = int(gen.integers(1, 5)), int(gen.choice([1, 1, 2, 3]))
        m = n
        n = m * fpa
        fr = torch.from_numpy(gen.standard_normal((n, 3, hh, ww), dtype=np.float32))
        act = torch.from_numpy(gen.random((m, 2), dtype=np.float32))
        sizes = torch.from_numpy(gen.integers(1, hh + 1, size=(m,)).astype(np.int32))
        if trial % 3 == 0:
            sizes[0] = p
        layout = (ops.LAYOUT_NCHW, ops.LAYOUT_NHWC, ops.LAYOUT_NHWC4)[trial % 3]
        got = ops.crop_resize(fr.to(dev), act.to(dev), p, size=sizes.to(dev), frames_per_action=fpa, layout=layout)
        got = _to_nchw(got, layout, ops).cpu()
        ref = O.crop_resize(fr, act.repeat_interleave(fpa, 0), sizes.repeat_interleave(fpa).tolist(), p)
        assert (got - ref).abs().max().item() <= 2e-6, (trial, hh, ww, p, fpa)
    # argument errors
    from adafocus_amd._lib import AdafError
    fr = torch.zeros((2, 3, 64, 64), device=dev)
    act = torch.zeros((2, 2), device=dev)
    with pytest.raises(AdafError):
        ops.crop_resize(fr, act, 32, size=65)
    with pytest.raises(AdafError):
        ops.crop_resize(fr, act, 32, size=0)


def test_crop_resize_full_size_properties(dev, ops):
    """BASELINE's full size (1024 frames of 224^2): (i) window size == patch size is the gather, bit for bit, through the
    resampling kernel; (ii) a resample of a per-frame constant image is that constant (the four weights sum to 1);
    (iii) resampling commutes with a per-frame affine map of the pixel values (linearity), to fp32 rounding."""
    n, p = 1024, 96
    gen = torch.Generator(device="cpu").manual_seed(77)
    fr = torch.randn((n, 3, 224, 224), generator=gen).to(dev)
    act = torch.rand((n, 2), generator=gen).to(dev)
    same = torch.full((n,), p, dtype=torch.int32, device=dev)
    assert torch.equal(ops.crop_resize(fr, act, p, size=same, layout=ops.LAYOUT_NHWC4), ops.crop_gather(fr, act, p, 1, ops.LAYOUT_NHWC4))
    sizes = torch.randint(32, 225, (n,), generator=gen, dtype=torch.int32).to(dev)
    consts = torch.randn((n, 1, 1, 1), generator=gen).to(dev)
    flat = consts.expand(n, 3, 224, 224).contiguous()
    out = ops.crop_resize(flat, act, p, size=sizes)
    assert (out - consts).abs().max().item() <= 1e-6 * max(1.0, float(consts.abs().max()))
    base = ops.crop_resize(fr, act, p, size=sizes)
    a_, b_ = 1.75, -0.5
    lin = ops.crop_resize(fr * a_ + b_, act, p, size=sizes)
    assert (lin - (base * a_ + b_)).abs().max().item() < 2e-5


def test_resize_nearest_golden(dev, ops):
    g = golden("g10_resample")
    fr2 = rnd((2, 6, 224, 224), 102)
    for gs in (160, 128, 112, 96):
        o = ops.resize_nearest(fr2.to(dev), gs).cpu().numpy()
        assert o.shape == (2, 6, gs, gs)
        assert np.array_equal(_sha(o), g["nearest_sha_%d" % gs])
    fr = rnd((3, 3, 224, 224), 103).to(dev)
    ref = torch.nn.functional.interpolate(fr.cpu(), (100, 100))
    o4 = ops.resize_nearest(fr, 100, ops.LAYOUT_NHWC4)
    assert torch.equal(o4[..., :3].permute(0, 3, 1, 2).cpu(), ref) and float(o4[..., 3].abs().max()) == 0.0
    fr4 = ops.crop_gather(fr, torch.zeros((3, 2), device=dev), 224, 1, ops.LAYOUT_NHWC4)
    assert torch.equal(ops.resize_nearest(fr4, 100, ops.LAYOUT_NHWC4), o4)


# ------------------------------------------------------------------------------------ models
def _act_args(**over):
    class A:
        pass
    a = A()
    a.__dict__.update(num_segments=8, num_classes=200, reward="random", dataset="actnet", input_size=224, batch_size=2,
                      patch_size=96, with_glancer=True, feature_map_channels=1280, glance_size=224, action_dim=49,
                      hidden_state_dim=1024, policy_conv=True, gpu=0, continuous=False, gamma=0.7, policy_lr=0.0003,
                      random_patch=False, dropout=0.5, consensus="gru", hidden_dim=1024)
    a.__dict__.update(over)
    return a


def _act_model(dev, **over):
    from adafocus_amd.gfv_net import GFV
    m = GFV(_act_args(**over)).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1007).items()}
    m.load_state_dict(sd, strict=True)
    return m.to(dev), sd


def test_glance_size_differs_from_input_size(dev, O):
    """glance_size = 160: the drivers feed the glancer F.interpolate(images, (160, 160)) (nearest; main_dist.py:331-332) and the
    policy's Linear is sized from ceil(160/32)^2 = 25 cells.  validate() and both offline forwards must follow."""
    from adafocus_amd import evaluate as E
    m, sd = _act_model(dev, glance_size=160)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=21))
    forced_idx, _ = synth.synth_actions(16, 7, seed=22)
    forced = torch.from_numpy(forced_idx).view(2, 8)
    with torch.no_grad():
        scan = O.glancer_input(frames, 160)
        ref_logits, ref_last, ref_idx, _, ref_gap = O.act_forward(sd, frames, scan, 96, 49, per_step=False, return_aux=True,
                                                                  return_gap=True)
        ref_f, ref_last_f = O.act_forward(sd, frames, scan, 96, 49, forced_action_idx=forced, per_step=False)
        x = frames.to(dev)
        assert torch.equal(m.glancer_input(x).cpu(), scan)
        lg_f, last_f, _, _ = m.offline_forward(x, m.glancer_input(x), forced)
        lg, last = m(input=x, scan=m.glancer_input(x), training=False, backbone_pred=False, one_step=True, gpu=0)
        _, _, _, idx = m.offline_forward(x, m.glancer_input(x))
    assert (lg_f.cpu() - ref_f).abs().max().item() < TOL and (last_f.cpu() - ref_last_f).abs().max().item() < TOL
    assert float(ref_gap.min()) >= 2e-3, "seed 21 puts the oracle's policy on an arg-max near-tie: pick another seed"
    assert torch.equal(idx.cpu(), ref_idx)
    assert (lg.cpu() - ref_logits).abs().max().item() < TOL

    labels = torch.tensor([[3], [150]], dtype=torch.int64)

    class DS:
        def __len__(self):
            return 2

        def __getitem__(self, i):
            return frames[i], labels[i]

    a = _act_args(glance_size=160)
    with torch.no_grad():
        r = E.validate(DS(), m, torch.nn.CrossEntropyLoss(), a, quiet=True)     # used to die in a .view()
    assert np.isfinite(r[0]) and np.isfinite(r[2])


def test_per_step_surface_golden(dev, ops):
    """a3: Focuser.forward / PatchSampler.sample / backbone_pred / LinearCLassifier with the reference's signatures."""
    from adafocus_amd.gfv_net import LinearCLassifier
    g = golden("g11_act_surface")
    m, _ = _act_model(dev)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=0)).to(dev)
    fr5 = frames.view(2, 8, 3, 224, 224)
    with torch.no_grad():
        fm, fv = m.glance(frames)
        assert fm.shape == (2, 8, 1280, 7, 7) and fv.shape == (2, 8, 1280)
        for s in range(3):
            feat, (none, std_action) = m.focuser(input=fr5[:, s], state=fm[:, s], restart_batch=(s == 0), training=False)
            assert none is None and feat.shape == (2, 2048, 1, 1)
            assert np.array_equal(std_action.cpu().numpy(), g["focuser_action_%d" % s]), s
            assert np.abs(feat.view(2, -1).cpu().numpy() - g["focuser_feat_%d" % s]).max() < TOL
        # ppo.py:68-79: the zero state appended at restart_batch, then one entry per step
        assert len(m.focuser.memory.hidden) == 4 and m.focuser.memory.hidden[-1].shape == (1, 2, 1024)
        assert float(m.focuser.memory.hidden[0].abs().max()) == 0.0
        a = torch.from_numpy(g["sample_action"]).to(dev)
        assert np.array_equal(_sha(m.focuser.patch_sampler.sample(fr5[:, 1].contiguous(), a).cpu().numpy()), g["sample_sha"])
        small = frames[:, :6].contiguous()
        pf = m(input=small, backbone_pred=True, glancer=False)
        pg = m(input=small, backbone_pred=True, glancer=True)
    assert pf.shape == (2, 2, 200) and np.abs(pf.cpu().numpy() - g["pred_focuser"]).max() < TOL
    assert np.abs(pg.cpu().numpy() - g["pred_glancer"]).max() < TOL
    lin = LinearCLassifier(seq_len=8, input_dim=3328, batch_size=2, hidden_dim=1024, num_classes=200, dropout=0.5).eval()
    lin.load_state_dict({k: torch.from_numpy(v) for k, v in
                         synth.synth_state_dict({"fc.weight": (200, 3328), "fc.bias": (200,)}, 707).items()})
    lin = lin.to(dev)
    with torch.no_grad():
        lg, avg = lin(rnd((2, 8, 3328), 71, 0.5).to(dev))
    assert np.abs(lg.cpu().numpy() - g["linear_log"]).max() < 1e-4
    assert np.abs(avg.cpu().numpy() - g["linear_avg"]).max() < 1e-6


def test_act_config3_shape_golden(dev, O):
    """BASELINE config 3's shape (T = 16, P = 128) through the whole forward and through hot_path, against the reference."""
    g = golden("g7_act_c3")
    m, sd = _act_model(dev, num_segments=16, patch_size=128)
    frames = torch.from_numpy(synth.synth_frames(2, 16, 224, seed=7))
    forced = torch.from_numpy(g["forced_idx"])
    with torch.no_grad():
        x = frames.to(dev)
        lg_f, last_f, feat, _ = m.offline_forward(x, x, forced)
        _, _, _, idx = m.offline_forward(x, x)
        table = torch.from_numpy(synth.grid_table(7))
        hp_logits, hp_last, _ = m.hot_path(x.view(32, 3, 224, 224), feat[:, :, :1280].contiguous(),
                                           table[forced.reshape(-1)].to(dev), 2, 16)
    assert np.abs(lg_f.cpu().numpy() - g["logits_forced"]).max() < TOL
    assert np.abs(last_f.cpu().numpy() - g["last_forced"]).max() < TOL
    assert torch.equal(hp_logits, lg_f) and torch.equal(hp_last, last_f)
    assert g["policy_argmax_gap"].min() >= 2e-3 and np.array_equal(idx.cpu().numpy(), g["policy_idx"])
    # a full-width batch of the same shape against the oracle on a few clips (the oracle needs ~1 s per clip here)
    b = 16
    fr = torch.from_numpy(synth.synth_frames(b, 16, 224, seed=8))
    fidx, act = synth.synth_actions(b * 16, 7, seed=9)
    gvec = rnd((b, 16, 1280), 81, 0.5)
    with torch.no_grad():
        lg, last, _ = m.hot_path(fr.view(b * 16, 3, 224, 224).to(dev), gvec.to(dev), torch.from_numpy(act).to(dev), b, 16)
        rl, rlast = O.act_hot_path(sd, fr.view(b * 16, 3, 224, 224)[:48], gvec[:3], torch.from_numpy(act)[:48], 128)
    assert (lg.cpu().view(b, 16, -1)[:3].reshape(48, -1) - rl).abs().max().item() < TOL
    assert (last.cpu()[:3] - rlast).abs().max().item() < TOL


def _sth_model(dev, vd):
    from adafocus_amd.gfv_net_sth import GFV
    from tests.test_state_dict_compat import sth_args
    a = sth_args()
    a.gpu, a.video_div = 0, vd
    m = GFV(a).eval()
    m.focuser.net.base_model = torch.nn.Sequential(*list(m.focuser.net.base_model.children())[:-1])  # evaluate.py:83
    m.load_state_dict(synth_sd("STH", 1007), strict=True)
    pol = {k[len("policy."):]: v for k, v in synth_sd("STH_POLICY" if vd == 1 else "STH_POLICY_VD2", 1007).items()}
    m.focuser.policy.policy_old.load_state_dict(pol)
    m.focuser.policy.policy.load_state_dict(pol)
    m.focuser.policy.policy_old.eval()
    m.focuser.policy.policy.eval()
    return m.to(dev), a


@pytest.mark.parametrize("vd", [1, 2])
def test_sth_video_div_and_baseline_golden(dev, vd):
    """STH/evaluate.py:198-210 loop for video_div = 1 and 2: the policy's GRU state and the previous steps' patches are
    carried, and the reward-baseline logits (random_patching with the reference's recorded torch.rand draw injected) are
    checked by VALUE against the reference."""
    g = golden("g12_sth_steps")
    m, a = _sth_model(dev, vd)
    gl = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=3)).to(dev)
    fo = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=4)).view(2, 8, 3, 224, 224).to(dev)
    with torch.no_grad():
        fm, glog = m.glance(gl)
        prev = None
        for step in range(vd):
            rand = torch.from_numpy(g["vd%d_rand_%d" % (vd, step)]).to(dev)
            total, base, patch = m.action_stage2(fo, fm, glog, step, a, prev_local_patch=prev, training=False, baseline_action=rand)
            hid = m.focuser.memory.hidden[-1]
            assert hid.shape == (1, 2, 1024) and len(m.focuser.memory.hidden) == step + 2      # zero state + one per step
            assert np.abs(hid[0].cpu().numpy() - g["vd%d_hidden_%d" % (vd, step)]).max() < 1e-3
            assert g["vd%d_action_px_margin_%d" % (vd, step)].min() >= 0.02       # the reference's own margin: no escape hatch
            assert np.array_equal(patch[:, :, :, :4, :4].cpu().numpy(), g["vd%d_patch_corner_%d" % (vd, step)])
            assert np.abs(total.cpu().numpy() - g["vd%d_total_%d" % (vd, step)]).max() < TOL
            assert np.abs(base.cpu().numpy() - g["vd%d_base_%d" % (vd, step)]).max() < TOL
            if step == 0:      # stage 3's forward is the same main branch (gfv_net.py:190-225)
                total3, patch3 = m.action_stage3(fo, fm, glog, 0, a, prev_local_patch=None)
                assert torch.equal(total3, total) and torch.equal(patch3, patch)
            prev = patch


def test_sth_stage3_classifier_training_forward(dev, O):
    """f4: action_stage3 in stage-3 training mode (STH/stage3.py:313-317, 351-353: model.train() with glancer / focuser /
    policy in eval mode): the local CNN runs frozen on the HIP trunk, the loss back-propagates into classifier.weight /
    classifier.bias through two engine GEMMs; gradients against torch-CPU autograd on the oracle's features."""
    g = golden("g7_sth_e2e")
    m, a = _sth_model(dev, 1)
    m.dropout.p = 0.0                       # dropout draws from the device RNG: compared with it disabled
    m.train()
    m.glancer.eval()
    m.focuser.eval()
    m.focuser.policy.policy.eval()
    m.focuser.policy.policy_old.eval()
    for p_ in m.parameters():
        p_.requires_grad_(False)
    for p_ in m.classifier.parameters():
        p_.requires_grad_(True)
    gl = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=3)).to(dev)
    fo = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=4)).view(2, 8, 3, 224, 224).to(dev)
    forced = torch.from_numpy(g["forced_action"]).to(dev)
    target = torch.tensor([5, 170], device=dev)
    with torch.no_grad():
        fm, glog = m.glance(gl)
    pred, patch = m.action_stage3(fo, fm, glog, 0, a, prev_local_patch=None, forced_action=forced)
    assert pred.requires_grad and not patch.requires_grad
    assert np.abs(pred.detach().cpu().numpy() - g["logits_stage3_forced"]).max() < TOL
    loss = torch.nn.functional.cross_entropy(pred, target)
    loss.backward()
    gw, gb = m.classifier.weight.grad.cpu(), m.classifier.bias.grad.cpu()
    # oracle: same features on the CPU, autograd through Linear + mean
    sd = synth_sd("STH", 1007)
    sd.update(synth_sd("STH_POLICY", 1007))
    sd = O.canonical_resnet_keys(sd, "focuser.net.base_model.")
    with torch.no_grad():
        feat = O.resnet50_trunk(sd, "focuser.net.base_model.", patch.detach().cpu().reshape(16, 3, 128, 128), 8, 8).flatten(1)
    w = sd["classifier.weight"].clone().requires_grad_(True)
    bb = sd["classifier.bias"].clone().requires_grad_(True)
    ref = torch.nn.functional.linear(feat, w, bb).view(2, 8, -1).mean(1) + glog.detach().cpu().mean(1)
    torch.nn.functional.cross_entropy(ref, target.cpu()).backward()
    assert (gw - w.grad).abs().max().item() < 1e-4 and (gb - bb.grad).abs().max().item() < 1e-4
    assert gw.abs().max().item() > 1e-3            # a real gradient
    # one SGD step moves the logits the way the CPU model's step does
    with torch.no_grad():
        m.classifier.weight -= 0.1 * m.classifier.weight.grad
        m.classifier.bias -= 0.1 * m.classifier.bias.grad
        w2, b2 = w - 0.1 * w.grad, bb - 0.1 * bb.grad
    pred2, _ = m.action_stage3(fo, fm, glog, 0, a, prev_local_patch=None, forced_action=forced)
    ref2 = torch.nn.functional.linear(feat, w2, b2).view(2, 8, -1).mean(1) + glog.detach().cpu().mean(1)
    assert (pred2.detach().cpu() - ref2.detach()).abs().max().item() < TOL * max(1.0, ref2.detach().abs().max().item())
    assert (pred2.detach() - pred.detach()).abs().max().item() > 1e-3          # the step really changed the logits
    m.eval()


# ------------------------------------------------------------------------------------ fused trunk launches
@pytest.mark.parametrize("p,n,tsm", [(96, 8, 0), (128, 4, 0), (100, 3, 0), (72, 4, 4), (96, 16, 8), (64, 1, 0), (33, 5, 0),
                                     # >= 128 patches: the fused tails run position-major tiles (full groups, a ragged last group, no next conv1)
                                     (32, 128, 0), (40, 250, 0), (32, 256, 8),
                                     # round 5: at >= 256 patches of an instantiated size the fused stem is the strip-walking kernel
                                     (64, 257, 0), (96, 264, 12), (128, 256, 0),
                                     # round 6: a shifted next conv1 rides in the position-major fused tail when clips divide the 128-image tiles
                                     # (full groups, three groups, a ragged last group whose last valid frame ends a clip)
                                     (96, 256, 16), (48, 384, 8), (32, 248, 8),
                                     # ... and with 120-image tile groups when clips of 12 frames do not divide 128
                                     (48, 360, 12), (32, 240, 12)])
def test_resnet50_fused_launches_bit_identical(dev, p, n, tsm):
    """Stage-1 conv2 -> conv3 -> next conv1 in one launch and stem + max-pool in one launch (adaf_resnet50_set_fusion):
    same k order in every product, hence torch.equal with the one-launch-per-layer plan -- full tiles, ragged last tiles
    (n * (p/4)^2 not a multiple of 128), with the temporal shift (next conv1 left out) and without; row-major tiles (few
    patches) and position-major tiles with tap skipping (>= 128 patches)."""
    from adafocus_amd.resnet import resnet50
    net = resnet50(num_classes=10).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 11).items()})
    net = net.to(dev)
    net.set_math("f32")
    net.tsm_segments = tsm
    x = rnd((n, p, p, 4), 900 + p + n).to(dev)
    x[..., 3] = 0
    net.set_fusion(False)
    ref = net.features_nhwc4(x).clone()
    prof_ref = net._sync().profile(x, tsm_segments=tsm)
    net.set_fusion(2)          # 2: the fused stem launch at every patch size, not only where it is the faster plan
    got = net.features_nhwc4(x).clone()
    prof = net._sync().profile(x, tsm_segments=tsm)
    net.set_fusion(True)
    assert torch.equal(net.features_nhwc4(x), ref)
    assert torch.isfinite(ref).all()
    assert torch.equal(got, ref)
    assert len(prof) < len(prof_ref)                       # the fused plan really ran (fewer launches)
    if tsm:                # ... with the shifted next conv1 inside the tail (tile id 92) where whole clips fill position-major tile groups, else without
        gs = (128 // tsm) * tsm
        groups = (n + gs - 1) // gs
        rides = n >= 128 and groups * 128 * 100 <= n * (106 if gs == 128 else 108)
        assert sum(e["tile"] == 92 for e in prof) == (3 if rides else 0), (rides, [e["tile"] for e in prof][:8])
    assert abs(sum(e["flops"] for e in prof) - sum(e["flops"] for e in prof_ref)) < 1e-6 * sum(e["flops"] for e in prof_ref)


def test_conv_position_major_tap_skipping_bit_identical(dev, ops):
    """k x k convs on small maps run position-major tiles that skip the filter taps lying wholly in the padding
    (adaf_set_conv_pos_major): bit-identical to the row-major tiles, for every tile shape, strides 1 / 2, maps from 1x1 to
    12x12, image counts that fill, overfill and underfill a tile group, with and without a residual."""
    gen = np.random.Generator(np.random.PCG64([61, 7]))
    cases = [(256, 3, 3, 64, 64, 3, 1, 1), (130, 6, 6, 32, 96, 3, 1, 1), (128, 6, 6, 64, 32, 3, 2, 1), (64, 3, 3, 128, 64, 3, 1, 1),
             (200, 1, 1, 32, 64, 3, 1, 1), (129, 12, 12, 32, 64, 3, 1, 1), (70, 5, 7, 32, 64, 3, 2, 1), (256, 2, 2, 64, 128, 5, 1, 2)]
    try:
        for n, hh, ww, cin, cout, k, stride, pad in cases:
            x = torch.from_numpy(gen.standard_normal((n, hh, ww, cin), dtype=np.float32)).to(dev)
            w = ops.pack_conv_weight(torch.from_numpy(gen.standard_normal((cout, cin, k, k), dtype=np.float32) * np.float32(0.05)).to(dev))
            sc = torch.from_numpy(gen.random(cout, dtype=np.float32) + np.float32(0.5)).to(dev)
            bi = torch.from_numpy(gen.standard_normal(cout, dtype=np.float32)).to(dev)
            oh, ow = (hh + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1
            res = torch.from_numpy(gen.standard_normal((n, oh, ow, cout), dtype=np.float32)).to(dev)
            for tile in (0, 31, 32, 33, 34):
                for r in (None, res):
                    ops.set_conv_pos_major(False, dev)
                    ref = ops.conv2d_bn_act(x, w, sc, bi, r, stride=stride, pad=pad, act=ops.ACT_RELU, tile=tile).clone()
                    ops.set_conv_pos_major(True, dev)
                    got = ops.conv2d_bn_act(x, w, sc, bi, r, stride=stride, pad=pad, act=ops.ACT_RELU, tile=tile)
                    assert torch.equal(got, ref), (n, hh, ww, cin, cout, k, stride, tile, r is not None)
            naive = ops.conv2d_bn_act(x, w, sc, bi, res, stride=stride, pad=pad, act=ops.ACT_RELU, naive=True)
            assert (got - naive).abs().max().item() < 2e-4
    finally:
        ops.set_conv_pos_major(True, dev)


# ------------------------------------------------------------------------------------ N2: half-precision storage
@pytest.mark.parametrize("tile", [0, 81, 82, 83, 84, 88])
def test_conv_f16_operands_vs_fp32_reference(dev, ops, tile):
    """adaf_conv2d_bn_act_f16: fp16 x / w, fp32 accumulate.  With operands that ARE fp16 values the products are exact, so
    against an fp32 conv of the same values only the summation order (fp32 output) or one fp16 rounding (fp16 output) differs."""
    gen = np.random.Generator(np.random.PCG64([41, tile]))
    shapes = [(6, 14, 14, 64, 384, 1, 1, 0, False), (3, 7, 7, 960, 160, 1, 1, 0, True), (2, 28, 28, 24, 144, 1, 1, 0, False),
              (5, 9, 9, 128, 64, 3, 2, 1, False), (2, 12, 12, 64, 64, 3, 1, 1, True), (1, 1, 1, 1280, 200, 1, 1, 0, False)]
    for n, hh, ww, cin, cout, k, stride, pad, with_res in shapes:
        x = torch.from_numpy(gen.standard_normal((n, hh, ww, cin), dtype=np.float32)).half()
        w = torch.from_numpy(gen.standard_normal((cout, cin, k, k), dtype=np.float32) * np.float32(1.0 / np.sqrt(cin * k * k))).half()
        sc = torch.from_numpy(gen.random(cout, dtype=np.float32) + np.float32(0.5))
        bi = torch.from_numpy(gen.standard_normal(cout, dtype=np.float32))
        oh = (hh + 2 * pad - k) // stride + 1
        res = torch.from_numpy(gen.standard_normal((n, oh, oh, cout), dtype=np.float32)).half() if with_res else None
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), stride=stride, padding=pad)
        ref = ref * sc[None, :, None, None] + bi[None, :, None, None]
        if res is not None:
            ref = ref + res.float().permute(0, 3, 1, 2)
        ref = torch.relu(ref).permute(0, 2, 3, 1)
        w16 = ops.pack_conv_weight_f16(w.float().to(dev))
        assert torch.equal(w16.cpu(), w.permute(0, 2, 3, 1).contiguous())          # fp16 values survive the packer exactly
        for odt in (torch.float32, torch.float16):
            got = ops.conv2d_bn_act_f16(x.to(dev), w16, sc.to(dev), bi.to(dev), None if res is None else res.to(dev), stride=stride,
                                        pad=pad, act=ops.ACT_RELU, out_dtype=odt, tile=tile)
            assert got.dtype == odt
            err = (got.float().cpu() - ref).abs().max().item()
            assert err < (2e-4 if odt == torch.float32 else 2e-3 * max(1.0, ref.abs().max().item())), (cin, cout, k, odt, err)


def test_dwconv_f16_and_casts(dev, ops):
    gen = np.random.Generator(np.random.PCG64([42, 1]))
    for n, hh, c, stride in ((3, 14, 384, 1), (2, 15, 96, 2), (1, 7, 960, 1), (4, 1, 32, 1)):
        x = torch.from_numpy(gen.standard_normal((n, hh, hh, c), dtype=np.float32)).half()
        w = torch.from_numpy(gen.standard_normal((c, 1, 3, 3), dtype=np.float32) * np.float32(0.3))
        sc = torch.from_numpy(gen.random(c, dtype=np.float32) + np.float32(0.5))
        bi = torch.from_numpy(gen.standard_normal(c, dtype=np.float32))
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, stride=stride, padding=1, groups=c)
        ref = torch.clamp(ref * sc[None, :, None, None] + bi[None, :, None, None], 0, 6).permute(0, 2, 3, 1)
        got = ops.dwconv3x3_bn_act_f16(x.to(dev), ops.pack_dw_weight(w.to(dev)), sc.to(dev), bi.to(dev), stride=stride)
        assert got.dtype == torch.float16 and (got.float().cpu() - ref).abs().max().item() < 4e-3
    v = torch.from_numpy(gen.standard_normal(1001, dtype=np.float32) * 100).to(dev)
    assert torch.equal(ops.cast(v, torch.float16).cpu(), v.cpu().half())
    assert torch.equal(ops.cast(ops.cast(v, torch.float16), torch.float32).cpu(), v.cpu().half().float())


def test_mobilenetv2_blocks_from_f16_entry_points_vs_g5_golden(dev, ops):
    """Single MobileNetV2 blocks of G5 (generated by the real reference in fp32) assembled from the fp16-storage entry points
    (adaf_conv2d_bn_act_f16, adaf_dwconv3x3_bn_act_f16: the building blocks config 5 runs on).  The glancer itself has no fp16 mode any
    more (round 6: adaf_mobilenetv2_set_dtype measured 1.01x and cost 15 % of the policy's arg-max choices)."""
    from adafocus_amd.mobilenet import mobilenet_v2
    g = golden("g5_mbv2_act")
    mb = mobilenet_v2().eval()
    shapes = {k: tuple(v.shape) for k, v in mb.state_dict().items()}
    mb.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 505).items()})
    mb = mb.to(dev)
    xc = rnd((2, 3, 64, 64), 53).to(dev)
    fm32, fv32 = mb.features_nhwc(xc)
    ref_fm = torch.from_numpy(g["fm"]).permute(0, 2, 3, 1)
    assert (fm32.cpu() - ref_fm).abs().max().item() < 1e-3
    # single inverted-residual blocks (t = 6, stride 1 with identity; t = 6, stride 2) assembled from the fp16 entry points
    sd = {k: v.to(dev) for k, v in mb.state_dict().items()}
    xb = rnd((2, 24, 16, 16), 52)

    def block(i, stride, residual):
        pfx = "features.%d.conv." % i
        x16 = xb.permute(0, 2, 3, 1).contiguous().half().to(dev)
        we = ops.pack_conv_weight_f16(sd[pfx + "0.0.weight"])
        se, be = ops.fold_bn(sd[pfx + "0.1.weight"], sd[pfx + "0.1.bias"], sd[pfx + "0.1.running_mean"], sd[pfx + "0.1.running_var"])
        e = ops.conv2d_bn_act_f16(x16, we, se, be, act=ops.ACT_RELU6)
        sdw, bdw = ops.fold_bn(sd[pfx + "1.1.weight"], sd[pfx + "1.1.bias"], sd[pfx + "1.1.running_mean"], sd[pfx + "1.1.running_var"])
        d = ops.dwconv3x3_bn_act_f16(e, ops.pack_dw_weight(sd[pfx + "1.0.weight"]), sdw, bdw, stride=stride)
        wp = ops.pack_conv_weight_f16(sd[pfx + "2.weight"])
        sp, bp = ops.fold_bn(sd[pfx + "3.weight"], sd[pfx + "3.bias"], sd[pfx + "3.running_mean"], sd[pfx + "3.running_var"])
        return ops.conv2d_bn_act_f16(d, wp, sp, bp, x16 if residual else None, act=ops.ACT_NONE, out_dtype=torch.float32)
    for name, i, stride, residual in (("r2", 3, 1, True), ("r3", 4, 2, False)):
        ref = torch.from_numpy(g[name]).permute(0, 2, 3, 1)
        got = block(i, stride, residual).cpu()
        assert (got - ref).abs().max().item() < 2e-2 * max(1.0, float(ref.abs().max())), name


@pytest.mark.parametrize("vd", [1, 2])
def test_validate_sth_loop_uint8_clips_equal_fp32_clips(dev, vd, O):
    """The Something-Something loop fed with the loader's stacked uint8 clips ((H, W, T*3) per stream: what Stack() hands
    ToTorchFormatTensor, STH/ops/transforms.py:303-336) -- normalised on the GPU, both streams staged in their own pinned
    buffers, the glancer and the patch gather reading the pixel-major frames directly -- must give the logits of the same
    clips normalised on the host as the reference does (GroupNormalize, :64-77): torch.equal, with the baseline branch
    (same torch.rand stream) and without, for video_div = 1 (ragged last batch) and 2 (whole clip pairs: the temporal shift views
    b * Tf / video_div frames as clips of Tf, so a partial pass needs an even batch -- in the reference too)."""
    from adafocus_amd import evaluate as E
    m, a = _sth_model(dev, vd)
    a.batch_size, a.glance_size = 2, 224
    gen = np.random.Generator(np.random.PCG64([19, vd]))
    n = 5 if vd == 1 else 6
    gu = gen.integers(0, 256, size=(n, 224, 224, 24), dtype=np.uint8)
    fu = gen.integers(0, 256, size=(n, 224, 224, 24), dtype=np.uint8)
    gf = torch.stack([O.ingest_uint8(gu[i]) for i in range(n)])
    ff = torch.stack([O.ingest_uint8(fu[i]) for i in range(n)])
    labels = torch.tensor([5, 100, 3, 77, 173, 9])[:n]

    class DS:
        def __init__(self, g, f):
            self.g, self.f = g, f

        def __len__(self):
            return n

        def __getitem__(self, i):
            return self.g[i], self.f[i], labels[i]
    for wb in (True, False):
        torch.manual_seed(11)
        r8 = E.validate_sth(DS(torch.from_numpy(gu), torch.from_numpy(fu)), m, torch.nn.CrossEntropyLoss(), a, quiet=True,
                            with_baseline=wb, return_logits=True)
        torch.manual_seed(11)
        r32 = E.validate_sth(DS(gf, ff), m, torch.nn.CrossEntropyLoss(), a, quiet=True, with_baseline=wb, return_logits=True)
        assert torch.equal(r8[4], r32[4]) and torch.equal(r8[5], r32[5])
        assert r8[:3] == r32[:3], (r8[:3], r32[:3])


@pytest.mark.parametrize("vd", [1, 2])
def test_validate_sth_loop_against_golden(dev, vd):
    """evaluate.validate_sth = the loop of STH/evaluate.py:165-226 (two frame streams, video_div focusing steps, baseline
    branch + reward bookkeeping) on the HIP model: the last step's logits equal the reference's (G12), the metrics are
    computed over the gathered set, the rewards are finite and logged per step."""
    from adafocus_amd import evaluate as E
    g = golden("g12_sth_steps")
    m, a = _sth_model(dev, vd)
    a.batch_size, a.glance_size = 2, 224
    gl = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=3))
    fo = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=4))
    labels = torch.tensor([5, 100])

    class DS:
        def __len__(self):
            return 2

        def __getitem__(self, i):
            return gl[i], fo[i], labels[i]
    t1, t5, rew, logs, lg, tg = E.validate_sth(DS(), m, torch.nn.CrossEntropyLoss(), a, quiet=True, return_logits=True)
    assert np.abs(m.focuser.memory.hidden[-1][0].cpu().numpy() - g["vd%d_hidden_%d" % (vd, vd - 1)]).max() < 1e-3   # the policy's carried state
    assert np.abs(lg.numpy() - g["vd%d_total_%d" % (vd, vd - 1)]).max() < TOL
    assert torch.equal(tg, labels) and len(rew) == vd and all(np.isfinite(r) for r in rew)
    ref1 = float(E.accuracy(torch.from_numpy(g["vd%d_total_%d" % (vd, vd - 1)]), labels)[0])
    assert abs(t1 - ref1) < 1e-4 and logs[-1].startswith(" * Acc@1")
    # the baseline branch is optional (it only feeds the logged reward): same logits without it
    nb = E.validate_sth(DS(), m, torch.nn.CrossEntropyLoss(), a, quiet=True, with_baseline=False, return_logits=True)
    assert torch.equal(nb[4], lg) and nb[2] == [None] * vd


def test_latency_mode_graph_replay_bit_identical(dev, ops):
    """GFV.capture_hot_path: BASELINE config 1's step (B = 2, T = 8, P = 96) captured into a HIP graph and replayed on new
    inputs equals the eagerly launched step bit for bit (same kernels, same order), also for B = 1 -- the default capture (GRU in
    its launch-per-step form: no grid barrier inside a graph) against eager mode 0, the `exclusive` capture (persistent scan in the
    graph, time-out counter checked after every replay) against the default eager step."""
    m, _ = _act_model(dev)
    for b in (2, 1):
        t = 8
        g = m.capture_hot_path(b, t)
        gx = m.capture_hot_path(b, t, exclusive=True, check_every=1)
        for seed in (61, 62, 63):
            fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=seed)).to(dev).view(b * t, 3, 224, 224)
            _, act = synth.synth_actions(b * t, 7, seed=seed + 10)
            act = torch.from_numpy(act).to(dev)
            gv = rnd((b, t, 1280), seed + 20, 0.5).to(dev)
            with torch.no_grad():
                lg, last, _ = m.hot_path(fr, gv, act, b, t)
                lg, last = lg.clone(), last.clone()
                try:
                    ops.set_gru_persistent(0, dev)
                    lg0, last0, _ = m.hot_path(fr, gv, act, b, t)
                    lg0, last0 = lg0.clone(), last0.clone()
                finally:
                    ops.set_gru_persistent(1, dev)
                glg, glast = [v.clone() for v in g(fr, gv, act)]
                xlg, xlast = [v.clone() for v in gx(fr, gv, act)]
            torch.cuda.synchronize()
            assert torch.equal(glg, lg0) and torch.equal(glast, last0), (b, seed)
            assert torch.equal(xlg, lg) and torch.equal(xlast, last), (b, seed)
            assert (lg0 - lg).abs().max().item() < 1e-4        # (the two GRU forms differ in summation order only)


def test_two_captured_hot_paths_replayed_concurrently_never_hand_out_nan(dev, ops):
    """two captured hot paths replayed at the same time on two streams.  Default captures carry no grid barrier, so
    every replay is bit-identical to the graph's own single-stream result.  `exclusive` captures (persistent scan, outside the slot
    throttle) either complete bit-identically or raise AdafError at their check -- NaN logits are never handed out silently."""
    from adafocus_amd._lib import AdafError
    m, _ = _act_model(dev)
    b, t = 2, 8
    fr = torch.from_numpy(synth.synth_frames(b, t, 224, seed=71)).to(dev).view(b * t, 3, 224, 224)
    act = torch.from_numpy(synth.synth_actions(b * t, 7, seed=72)[1]).to(dev)
    gv = rnd((b, t, 1280), 73, 0.5).to(dev)
    for exclusive in (False, True):
        g1 = m.capture_hot_path(b, t, exclusive=exclusive, check_every=1)
        g2 = m.capture_hot_path(b, t, exclusive=exclusive, check_every=1)
        with torch.no_grad():
            ref = g1(fr, gv, act)[0].clone()
            g2(fr, gv, act)
            torch.cuda.synchronize()
            s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            raised = False
            try:
                for _ in range(20):
                    with torch.cuda.stream(s1):
                        g1.graph.replay()
                    with torch.cuda.stream(s2):
                        g2.graph.replay()
                torch.cuda.synchronize()
                g1.check()
                g2.check()
            except AdafError:
                raised = True
        torch.cuda.synchronize()
        if exclusive:
            assert raised or (torch.equal(g1.logits, ref) and torch.equal(g2.logits, ref))
            assert raised or bool(torch.isfinite(g1.logits).all() and torch.isfinite(g2.logits).all())
        else:
            assert not raised and torch.equal(g1.logits, ref) and torch.equal(g2.logits, ref)
        del g1, g2


# ------------------------------------------------------------------------------------ GRU scan
def _gru_weights(dev):
    sd = synth_sd("ACT", 606, "classifier.", keep_prefix=False)
    d = {k: v.to(dev) for k, v in sd.items()}
    return sd, d


@pytest.mark.parametrize("b,t", [(1, 1), (5, 3), (64, 16), (65, 4), (96, 8), (128, 16), (200, 2)])
def test_gru_cls_scan_with_folded_fc(dev, ops, O, b, t):
    """One persistent kernel = recurrence + per-step nn.Linear + last-step rows, for batches up to 256 clips, against the
    launches-per-step form (mode 0), the cooperative launch (mode 2, bit-identical to mode 1) and the oracle."""
    sd, d = _gru_weights(dev)
    args = (d["gru.weight_ih_l0"], d["gru.weight_hh_l0"], d["gru.bias_ih_l0"], d["gru.bias_hh_l0"], d["fc.weight"], d["fc.bias"])
    x = rnd((b, t, 3328), 300 + b, 0.5)
    try:
        ops.set_gru_persistent(1, dev)
        l1, last1 = [v.clone() for v in ops.gru_cls_forward(x.to(dev), *args)]
        ops.set_gru_persistent(2, dev)
        l2, last2 = [v.clone() for v in ops.gru_cls_forward(x.to(dev), *args)]
        ops.set_gru_persistent(0, dev)
        l0, last0 = [v.clone() for v in ops.gru_cls_forward(x.to(dev), *args)]
    finally:
        ops.set_gru_persistent(1, dev)
    assert torch.isfinite(l1).all()
    assert torch.equal(l1, l2) and torch.equal(last1, last2)
    assert (l1 - l0).abs().max().item() < 5e-5 and (last1 - last0).abs().max().item() < 5e-5
    assert torch.equal(last1, l1.view(b, t, -1)[:, -1])
    if b <= 96:
        with torch.no_grad():
            rl, rlast = O.recurrent_classifier(sd, "", x)
        assert (l1.cpu() - rl).abs().max().item() < 1e-4 and (last1.cpu() - rlast).abs().max().item() < 1e-4


def test_gru_seq_with_initial_state(dev, ops, O):
    """h0: T = 1 with the previous call's state is one ActorCritic.act(restart_batch=False) step (ppo.py:70-79)."""
    sd, d = _gru_weights(dev)
    args = (d["gru.weight_ih_l0"][:, :1024].contiguous(), d["gru.weight_hh_l0"], d["gru.bias_ih_l0"], d["gru.bias_hh_l0"])
    cpu = [a.cpu() for a in args]
    for b in (2, 33, 70):
        x = rnd((b, 3, 1024), 400 + b, 0.5)
        h = torch.zeros(b, 1024)
        ref = []
        for s in range(3):
            h = O.gru_cell(x[:, s], h, *cpu)
            ref.append(h)
        for mode in (1, 0):
            try:
                ops.set_gru_persistent(mode, dev)
                whole = ops.gru_seq_forward(x.to(dev), *args)
                hs, steps = None, []
                for s in range(3):
                    hs = ops.gru_seq_forward(x[:, s:s + 1].to(dev), *args, h0=None if hs is None else hs.view(b, -1))
                    steps.append(hs.view(b, -1).clone())
            finally:
                ops.set_gru_persistent(1, dev)
            for s in range(3):
                assert (steps[s].cpu() - ref[s]).abs().max().item() < 1e-4, (b, mode, s)
                assert (whole[:, s].cpu() - ref[s]).abs().max().item() < 1e-4


def test_gru_scan_many_streams_with_foreign_kernels(dev, ops):
    """Scans from 8 streams interleaved with large GEMMs and small copy kernels (what an RCCL all-gather looks like to the
    scheduler): every result equals the single-stream one -- no scan was starved into its time-out / NaN path."""
    sd, d = _gru_weights(dev)
    args = (d["gru.weight_ih_l0"], d["gru.weight_hh_l0"], d["gru.bias_ih_l0"], d["gru.bias_hh_l0"], d["fc.weight"], d["fc.bias"])
    xs = [rnd((64, 16, 3328), 500 + i, 0.5).to(dev) for i in range(8)]
    refs = [ops.gru_cls_forward(x, *args)[0].clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    big = torch.randn(4096, 4096, device=dev)
    outs = [None] * 24
    for i in range(24):
        with torch.cuda.stream(streams[i % 8]):
            if i % 3 == 0:
                ops.linear(big, big)
            outs[i] = ops.gru_cls_forward(xs[i % 8], *args)[0]
            outs[i].clone()
    torch.cuda.synchronize()
    for i in range(24):
        assert torch.equal(outs[i], refs[i % 8]), i
    assert ops.gru_scan_timeouts(dev) == 0          # the device-side time-out counter the evaluation loop checks


# ------------------------------------------------------------------------------------ streams
def test_offline_forward_pipelined_equals_serial(dev, ops):
    """GFV.offline_forward_pipelined (front: ingest + glancer + policy, back: hot path, on the model's own streams) over
    consecutive different batches gives bit-identical results to the serial forward."""
    from adafocus_amd.transforms import ingest_uint8
    m, _ = _act_model(dev)
    gen = np.random.Generator(np.random.PCG64([31, 3]))
    clips = [torch.from_numpy(gen.integers(0, 256, size=(4, 224, 224, 24), dtype=np.uint8)).to(dev) for _ in range(5)]
    with torch.no_grad():
        serial = []
        for c in clips:
            lg, last, _, idx = m.offline_forward_nhwc4(ingest_uint8(c, 8), 4, 8)
            serial.append((lg.clone(), last.clone(), idx.clone()))
        torch.cuda.synchronize()
        piped = [m.offline_forward_pipelined(c, 8) for c in clips + clips]
        m.pipeline_flush()
        torch.cuda.synchronize()
    for i, (lg, last, idx, done, handoff) in enumerate(piped):
        ref = serial[i % 5]
        assert torch.equal(idx, ref[2]) and torch.equal(lg, ref[0]) and torch.equal(last, ref[1]), i


def test_pipelined_float_frames_release_event_is_done(dev):
    """with normalised float frames the back stream's gather reads the CALLER's buffer, so the release event
    returned as `handoff` must be `done`.  A slot that is overwritten as soon as `handoff` has passed (what evaluate.py
    does with its landing buffers) must not change the result."""
    m, _ = _act_model(dev)
    gen = np.random.Generator(np.random.PCG64([33, 3]))
    batches = [torch.from_numpy(gen.standard_normal((32, 224, 224, 4), dtype=np.float32)).to(dev) for _ in range(4)]
    for x in batches:
        x[..., 3] = 0
    with torch.no_grad():
        serial = [m.offline_forward_nhwc4(x, 4, 8)[0].clone() for x in batches]
        torch.cuda.synchronize()
        slot = torch.empty_like(batches[0])
        outs = []
        side = torch.cuda.Stream(device=dev)
        for i in range(8):
            slot.copy_(batches[i % 4])
            lg, last, idx, done, handoff = m.offline_forward_pipelined(slot, 8)
            assert handoff is done
            outs.append(lg)
            with torch.cuda.stream(side):          # the "loader": reuses the slot the moment the release event fires
                side.wait_event(handoff)
                slot.fill_(float("nan"))
            torch.cuda.current_stream().wait_stream(side)
        m.pipeline_flush()
        torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, serial[i % 4]), i


def test_policy_linear_encoder_and_hidden_list_like_the_reference(dev):
    """ActorCritic.act with policy_conv=False (ppo.py:40-47,75-76: the Linear encoder for ResNet / DenseNet features) on the
    engine, three steps, against the same nn.Modules on the CPU; memory.hidden holds the zero state + one entry per step."""
    from adafocus_amd.ppo import ActorCritic, Memory
    ac = ActorCritic(64, 64 * 7 * 7, 49, 1024, policy_conv=False).eval()
    shapes = {k: tuple(v.shape) for k, v in ac.state_dict().items()}
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 77).items()})
    states = [rnd((3, 64, 7, 7), 900 + i) for i in range(3)]
    with torch.no_grad():
        h = torch.zeros(1, 3, 1024)
        ref_h, ref_a = [], []
        for st in states:
            e = ac.state_encoder(st.flatten(1))
            o, h = ac.gru(e.view(1, 3, -1), h)
            ref_h.append(h.clone())
            ref_a.append(ac.actor(o[0]).max(1)[1])
    acd = ac.to(dev)
    mem = Memory()
    for i, st in enumerate(states):
        a = acd.act(st.to(dev), mem, restart_batch=(i == 0), training=False)
        assert len(mem.hidden) == i + 2 and float(mem.hidden[0].abs().max()) == 0.0
        assert (mem.hidden[-1].cpu() - ref_h[i]).abs().max().item() < 1e-4
        assert torch.equal(a.cpu(), ref_a[i])


def test_full_forward_on_three_streams_no_shared_scratch(dev):
    """the glancer's workspace is per stream.  Different batches enqueued round-robin on three streams (as
    bench.py's full-forward leg once did) must each equal their serial result."""
    m, _ = _act_model(dev)
    gen = np.random.Generator(np.random.PCG64([32, 3]))
    batches = [torch.from_numpy(gen.standard_normal((32, 224, 224, 4), dtype=np.float32)).to(dev) for _ in range(3)]
    for x in batches:
        x[..., 3] = 0
    with torch.no_grad():
        serial = [m.offline_forward_nhwc4(x, 4, 8)[0].clone() for x in batches]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        outs = []
        for i in range(12):
            with torch.cuda.stream(streams[i % 3]):
                outs.append(m.offline_forward_nhwc4(batches[i % 3], 4, 8)[0])
        torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, serial[i % 3]), i
    assert len(m.glancer.net._engine.sync()._scratch) >= 3


def test_scratch_cache_is_bounded(dev):
    """Short-lived streams must not pin a trunk workspace each (VERDICT r1 weak point): LRU of at most 6 streams."""
    from adafocus_amd.resnet import resnet50
    net = resnet50(num_classes=10).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 3).items()})
    net = net.to(dev)
    x = torch.randn((2, 64, 64, 4), device=dev)
    x[..., 3] = 0
    ref = net.features_nhwc4(x).clone()
    for _ in range(10):
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            out = net.features_nhwc4(x)
        s.synchronize()
        assert torch.equal(out, ref)
    assert len(net._sync()._scratch) <= 6


def test_device_mismatch_is_refused(dev, ops):
    """tensors that do not live on the current device are refused instead of launching on the wrong one."""
    if torch.cuda.device_count() < 2:
        from adafocus_amd import _lib
        x = torch.zeros(4, device=dev)
        _lib.on_current_device(x)          # same device: fine
        return
    from adafocus_amd._lib import AdafError
    x = torch.zeros((1, 3, 64, 64), device="cuda:1")
    with pytest.raises(AdafError):
        ops.crop_gather(x, torch.zeros((1, 2), device="cuda:1"), 32)




@pytest.mark.parametrize("rew", ["random", "prev"])
def test_one_step_act_validation_branch_golden(dev, rew):
    """GFV.one_step_act(training=False): the stage-2 validation loop body (ACT/main_dist.py:346-362, ACT/models/gfv_net.py:160-210,437-457) with
    the reference's per-step signature, against the reference's own outputs (G15, tools/gen_golden_r6.py).  reward = 'random': the baseline's
    crops are drawn from numpy's global generator exactly like the reference's (utils.py:31-32), so the same seed gives the same crops; 'prev':
    zeros beside the glancer vector.  The policy's arg-max margins are stored in the fixture (>= 2e-3), so actions compare unconditionally."""
    g = golden("g15_one_step_act")
    assert g["%s_policy_argmax_gap" % rew].min() >= 2e-3
    m, _ = _act_model(dev, reward=rew)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=5)).to(dev)
    fr5 = frames.view(2, 8, 3, 224, 224)
    state = np.random.get_state()
    np.random.seed(int(g["np_seed"][0]))
    try:
        with torch.no_grad():
            fm, fv = m.glance(frames)
            for s in range(int(g["steps"][0])):
                logits, last, psl, action, base = m.one_step_act(fr5[:, s], fm[:, s], fv[:, s], restart_batch=(s == 0), training=False)
                assert psl is None and logits.shape == (2, 200) and last.shape == (2, 200) and base.shape == (2, 200)
                assert np.array_equal(action.cpu().numpy(), g["%s_action_%d" % (rew, s)]), s
                assert np.abs(logits.cpu().numpy() - g["%s_logits_%d" % (rew, s)]).max() < TOL, s
                assert np.abs(last.cpu().numpy() - g["%s_last_%d" % (rew, s)]).max() < TOL, s
                assert np.abs(base.cpu().numpy() - g["%s_baseline_%d" % (rew, s)]).max() < TOL, s
            assert m.classifier.hx.shape == (1, 2, 1024)
            with pytest.raises(NotImplementedError):
                m.one_step_act(fr5[:, 0], fm[:, 0], fv[:, 0], restart_batch=True, training=True)
    finally:
        np.random.set_state(state)


def test_one_step_act_steps_equal_the_offline_forward(dev):
    """T one_step_act steps from restart_batch reproduce the batched offline forward's per-step logits (the restructuring of SURVEY §0.4
    changes the execution plan, not the result), and the classifier's step functions refuse a step without a state."""
    from adafocus_amd.gfv_net import RecurrentClassifier
    m, _ = _act_model(dev, reward="prev")
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=9)).to(dev)
    fr5 = frames.view(2, 8, 3, 224, 224)
    with torch.no_grad():
        ref_logits, ref_last = m.offline_forward(frames, frames)[:2]
        fm, fv = m.glance(frames)
        outs = []
        for s in range(8):
            logits, last, _, _, _ = m.one_step_act(fr5[:, s], fm[:, s], fv[:, s], restart_batch=(s == 0), training=False)
            outs.append(logits)
        assert (torch.stack(outs, 1).reshape(16, -1) - ref_logits).abs().max().item() < 1e-4
        assert (last - ref_last).abs().max().item() < 1e-4
    c = RecurrentClassifier(seq_len=8, input_dim=64, batch_size=2, hidden_dim=1024, num_classes=10, dropout=0.5).eval().to(dev)
    with pytest.raises(RuntimeError):
        c.single_forward(torch.zeros((2, 1, 64), device=dev), reset=False)


def test_validate_stage2_loop_on_the_hip_model(dev):
    """evaluate.validate with args.train_stage = 2 (ACT/main_dist.py:343-366): the MDP step by step through one_step_act on the HIP ops.  In eval
    mode the policy never sees local features, so the last step's prediction -- hence every metric -- equals the stage-3 loop's; the per-step
    mAP lines are there; glance_size != input_size goes through the nearest resize like the stage-3 branch."""
    from adafocus_amd import evaluate as E
    frames = torch.from_numpy(synth.synth_frames(3, 8, 224, seed=31))
    labels = torch.tensor([[3], [150], [42]], dtype=torch.int64)

    class DS:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            return frames[i], labels[i]
    for gs in (224, 160):
        m, _ = _act_model(dev, glance_size=gs, reward="prev")
        a3, a2 = _act_args(glance_size=gs), _act_args(glance_size=gs, train_stage=2)
        with torch.no_grad():
            r3 = E.validate(DS(), m, torch.nn.CrossEntropyLoss(), a3, quiet=True)
            r2 = E.validate(DS(), m, torch.nn.CrossEntropyLoss(), a2, quiet=True)
        assert r2[:3] == pytest.approx(r3[:3], abs=1e-4)
        assert sum(ln.startswith("mAP @ time step") for ln in r2[3]) == 8 and not any(ln.startswith("mAP @ time step") for ln in r3[3])


def test_stage1_form_in_eval_mode_golden(dev):
    """GFV.forward(one_step=False, training=False) -- the stage-1 form validate() runs at train_stage 1 (ACT/models/gfv_net.py:135-150,
    ACT/main_dist.py:334-340) -- against the reference (G15): a random_patch model with numpy seeded like the generator (same crops), and a
    policy model (ONE policy step over the B*T frames as a batch; margins stored in the fixture).  PatchSampler's random branch too."""
    g = golden("g15_one_step_act")
    assert g["s1_policy_argmax_gap"].min() >= 2e-3
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=int(g["s1_seed_frames"][0]))).to(dev)
    state = np.random.get_state()
    try:
        m, _ = _act_model(dev, random_patch=True)
        assert m.focuser.policy is None
        np.random.seed(int(g["np_seed"][0]) + 1)
        with torch.no_grad():
            lg, last = m(input=frames, scan=frames, training=False, backbone_pred=False, one_step=False)
        assert np.abs(lg.cpu().numpy() - g["s1_random_logits"]).max() < TOL and np.abs(last.cpu().numpy() - g["s1_random_last"]).max() < TOL
        # the sampler's random branch draws in the reference's order: same seed -> the recorded origins
        np.random.seed(int(g["np_seed"][0]) + 1)
        x = frames.view(16, 3, 224, 224)
        crops = m.focuser.patch_sampler.sample(x)
        for i, (y, xx) in enumerate(g["s1_random_origins"]):
            assert torch.equal(crops[i], x[i, :, y:y + 96, xx:xx + 96]), i
        assert m(input=frames, scan=frames, training=False, backbone_pred=False, one_step=True, gpu=0) is None      # (gfv_net.py:108: no body)
        with pytest.raises(NotImplementedError):
            m(input=frames, scan=frames, training=True, backbone_pred=False, one_step=False)
        m, _ = _act_model(dev)
        with torch.no_grad():
            lg, last = m(input=frames, scan=frames, training=False, backbone_pred=False, one_step=False)
        assert np.abs(lg.cpu().numpy() - g["s1_policy_logits"]).max() < TOL and np.abs(last.cpu().numpy() - g["s1_policy_last"]).max() < TOL
    finally:
        np.random.set_state(state)


def test_sth_patch_sampler_random_branch(dev):
    """STH/models/gfv_net.py:455-474: PatchSampler(random=True).sample crops every (B, 3T, H, W) clip at an origin drawn like the reference's
    random_crop (np.random.randint for y, then x)."""
    from adafocus_amd.gfv_net_sth import PatchSampler
    x = rnd((3, 12, 224, 224), 77).to(dev)
    ps = PatchSampler(128, True)
    state = np.random.get_state()
    try:
        np.random.seed(99)
        want = [(np.random.randint(0, 96), np.random.randint(0, 96)) for _ in range(3)]
        np.random.seed(99)
        got = ps.sample(x)
    finally:
        np.random.set_state(state)
    assert got.shape == (3, 12, 128, 128)
    for i, (y, xx) in enumerate(want):
        assert torch.equal(got[i], x[i, :, y:y + 128, xx:xx + 128]), i
