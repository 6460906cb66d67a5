"""Pin the round-2 additions of the oracle against vectors produced by the real reference
(tools/gen_golden_r2.py).  CPU only."""
import hashlib

import numpy as np
import torch

from adafocus_amd import synth
from oracle import ref_model as O
from tests.helpers import golden, rnd, synth_sd


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_g10_crop_resize_and_nearest():
    g = golden("g10_resample")
    fr = rnd((4, 3, 224, 224), 101)
    act = torch.from_numpy(g["actions"])
    for s_, p_ in g["cases"].tolist():
        o = O.crop_resize(fr, act, s_, p_).numpy()
        assert o.shape == (4, 3, p_, p_)
        np.testing.assert_allclose(o[:, :, ::7, ::5], g["sub_%d_%d" % (s_, p_)], rtol=0, atol=1e-6)
        np.testing.assert_allclose(o[:, :, -6:, -6:], g["corner_%d_%d" % (s_, p_)], rtol=0, atol=1e-6)
        if s_ == p_:    # scale 1: the resample IS the slice copy
            assert np.array_equal(_sha(o), g["sha_%d_%d" % (s_, p_)])
            assert np.array_equal(o, O.get_patch(fr, act, p_).numpy())
    o = O.crop_resize(fr, act, g["mixed_sizes"], 96).numpy()
    np.testing.assert_allclose(o[:, :, ::7, ::5], g["mixed_sub"], rtol=0, atol=1e-6)
    fr2 = rnd((2, 6, 224, 224), 102)
    for gs in (160, 128, 112, 96):
        assert np.array_equal(_sha(O.glancer_input(fr2, gs).numpy()), g["nearest_sha_%d" % gs])
    assert np.array_equal(O.glancer_input(fr2, 96).numpy()[:, :, -5:, -5:], g["nearest_corner_96"])


def test_g11_per_step_surface():
    g = golden("g11_act_surface")
    sd = synth_sd("ACT", 1007)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=0))
    fr5 = frames.view(2, 8, 3, 224, 224)
    table = O.standard_actions(49)
    with torch.no_grad():
        fm, _ = O.glancer_act(sd, "glancer.net.", frames.view(16, 3, 224, 224))
        fm = fm.view(2, 8, *fm.shape[1:])
        hid = frames.new_zeros(2, 1024)
        for s in range(3):      # Focuser.forward(input=, state=, restart_batch=s == 0, training=False), gfv_net.py:316-331
            idx, hid = O.policy_act_discrete(sd, "focuser.policy.policy_old.", fm[:, s], hid)
            assert np.array_equal(table[idx].numpy(), g["focuser_action_%d" % s])
            feat = O.resnet50_trunk(sd, "focuser.net.", O.get_patch(fr5[:, s], table[idx], 96)).view(2, -1)
            np.testing.assert_allclose(feat.numpy(), g["focuser_feat_%d" % s], rtol=1e-4, atol=2e-5)
        a = torch.from_numpy(g["sample_action"])
        assert np.array_equal(_sha(O.get_patch(fr5[:, 1], a, 96).numpy()), g["sample_sha"])     # PatchSampler.sample
        small = frames[:, :6].contiguous()
        np.testing.assert_allclose(O.backbone_pred(sd, small, "focuser").numpy(), g["pred_focuser"], rtol=1e-4, atol=5e-5)
        np.testing.assert_allclose(O.backbone_pred(sd, small, "glancer").numpy(), g["pred_glancer"], rtol=1e-4, atol=5e-5)
        lin_sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict({"fc.weight": (200, 3328), "fc.bias": (200,)}, 707).items()}
        lg, avg = O.linear_classifier(lin_sd, "", rnd((2, 8, 3328), 71, 0.5))
    np.testing.assert_allclose(lg.numpy(), g["linear_log"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(avg.numpy(), g["linear_avg"], rtol=1e-5, atol=1e-7)


def test_g7_act_config3_shape():
    g = golden("g7_act_c3")
    sd = synth_sd("ACT", 1007)
    frames = torch.from_numpy(synth.synth_frames(2, 16, 224, seed=7))
    with torch.no_grad():
        logits, last, idx, _ = O.act_forward(sd, frames, frames, 128, 49, per_step=False, return_aux=True)
        forced = torch.from_numpy(g["forced_idx"])
        logits_f, last_f = O.act_forward(sd, frames, frames, 128, 49, forced_action_idx=forced, per_step=False)
    assert np.array_equal(idx.numpy(), g["policy_idx"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logits_f.numpy(), g["logits_forced"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(last_f.numpy(), g["last_forced"], rtol=1e-4, atol=2e-5)
    assert len(set(g["forced_idx"].reshape(-1).tolist())) > 12


def sth_state(vd):
    sd = synth_sd("STH", 1007)
    sd.update(synth_sd("STH_POLICY" if vd == 1 else "STH_POLICY_VD2", 1007))
    return O.canonical_resnet_keys(sd, "focuser.net.base_model.")


def test_g12_sth_video_div_and_baseline():
    g = golden("g12_sth_steps")
    gl = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=3))
    fo = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=4)).view(2, 8, 3, 224, 224)
    for vd in (1, 2):
        sd = sth_state(vd)
        with torch.no_grad():
            fm, glog = O.glancer_sth(sd, "glancer.net.", gl.view(16, 3, 224, 224), 8, 8)
            fm, glog = fm.view(2, 8, *fm.shape[1:]), glog.view(2, 8, -1)
            hid, prev = None, None
            for step in range(vd):
                rand = torch.from_numpy(g["vd%d_rand_%d" % (vd, step)])
                total, base, prev, _, hid = O.sth_stage(sd, fm, glog, fo, step, vd, 128, 8, hid, prev, baseline_action=rand)
                np.testing.assert_allclose(hid.numpy(), g["vd%d_hidden_%d" % (vd, step)], rtol=1e-4, atol=1e-5)
                assert np.array_equal(prev[:, :, :, :4, :4].numpy(), g["vd%d_patch_corner_%d" % (vd, step)])
                np.testing.assert_allclose(total.numpy(), g["vd%d_total_%d" % (vd, step)], rtol=1e-4, atol=3e-5)
                np.testing.assert_allclose(base.numpy(), g["vd%d_base_%d" % (vd, step)], rtol=1e-4, atol=3e-5)


def test_g13_sth_shipped_configuration():
    """The reference's shipped Something-Something evaluation configuration (STH/evaluate.sh:5-15, conf/evaluate.yaml:29-30):
    Tg = 8 glancer frames, Tf = 12 focuser frames, P = 144.  The policy's state stays 1280 * 8 channels; the patches, the local
    CNN's batch and its temporal shift run over clips of TWELVE frames."""
    g = golden("g13_sth_shipped")
    sd = sth_state(1)
    gl = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=3))
    fo = torch.from_numpy(synth.synth_frames(2, 12, 224, seed=13)).view(2, 12, 3, 224, 224)
    with torch.no_grad():
        fm, glog = O.glancer_sth(sd, "glancer.net.", gl.view(16, 3, 224, 224), 8, 8)
        fm, glog = fm.view(2, 8, *fm.shape[1:]), glog.view(2, 8, -1)
        np.testing.assert_allclose(glog.numpy(), g["glancer_logit"], rtol=1e-4, atol=3e-5)
        total, base, patch, action, hid = O.sth_stage(sd, fm, glog, fo, 0, 1, 144, 12, None, None,
                                                      baseline_action=torch.from_numpy(g["rand"]))
        np.testing.assert_allclose(action.numpy(), g["policy_action"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(hid.numpy(), g["hidden"], rtol=1e-4, atol=1e-5)
        assert patch.shape == (2, 12, 3, 144, 144)
        assert np.array_equal(_sha(patch.numpy()), g["patch_sha"])
        assert np.array_equal(patch[:, :, :, :4, :4].numpy(), g["patch_corner"])
        np.testing.assert_allclose(total.numpy(), g["logits"], rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(total.numpy(), g["logits_stage3"], rtol=1e-4, atol=3e-5)     # stage 3 = the main branch
        np.testing.assert_allclose(base.numpy(), g["baseline"], rtol=1e-4, atol=3e-5)
        forced = torch.from_numpy(g["forced_action"])
        total_f, base_f, patch_f, _, _ = O.sth_stage(sd, fm, glog, fo, 0, 1, 144, 12, None, None, forced_action=forced,
                                                     baseline_action=torch.from_numpy(g["rand_forced"]))
        assert np.array_equal(_sha(patch_f.numpy()), g["patch_forced_sha"])
        np.testing.assert_allclose(total_f.numpy(), g["logits_forced"], rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(total_f.numpy(), g["logits_stage3_forced"], rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(base_f.numpy(), g["baseline_forced"], rtol=1e-4, atol=3e-5)
        # sth_forward (the evaluate.py:195-201 composition) at the same configuration
        t2, p2, a2 = O.sth_forward(sd, gl, fo, 144, 8, 12)
        assert torch.equal(p2, patch) and torch.equal(a2, action)
        np.testing.assert_allclose(t2.numpy(), g["logits"], rtol=1e-4, atol=3e-5)
    assert g["policy_action_px_margin"].min() >= 0.02       # the generator's floor: the GPU tests compare unconditionally


def test_policy_fixtures_carry_their_decision_margins():
    """Every policy-driven fixture stores how far the REFERENCE's policy output sat from a decision boundary (tools/gen_golden.py:
    ARGMAX_GAP_MIN, PIXEL_MARGIN_MIN), so no GPU test needs an escape hatch for an arg-max tie or a one-pixel crop move."""
    assert golden("g7_act_e2e")["policy_argmax_gap"].shape == (2, 8) and golden("g7_act_e2e")["policy_argmax_gap"].min() >= 2e-3
    assert golden("g7_act_c3")["policy_argmax_gap"].shape == (2, 16) and golden("g7_act_c3")["policy_argmax_gap"].min() >= 2e-3
    assert golden("g7_sth_e2e")["policy_action_px_margin"].min() >= 0.02
    g = golden("g12_sth_steps")
    for vd in (1, 2):
        for step in range(vd):
            assert g["vd%d_action_px_margin_%d" % (vd, step)].min() >= 0.02
    # the gap the fixtures store is the oracle's too (same policy arithmetic)
    sd = synth_sd("ACT", 1007)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=0))
    with torch.no_grad():
        fm, _ = O.glancer_act(sd, "glancer.net.", frames.view(16, 3, 224, 224))
        fm = fm.view(2, 8, *fm.shape[1:])
        hid = frames.new_zeros(2, 1024)
        gaps = []
        for s in range(8):
            _, hid, gap = O.policy_act_discrete(sd, "focuser.policy.policy_old.", fm[:, s], hid, return_gap=True)
            gaps.append(gap.numpy())
    np.testing.assert_allclose(np.stack(gaps, 1), golden("g7_act_e2e")["policy_argmax_gap"], rtol=0, atol=2e-4)


def _block_shift_tsn():
    from adafocus_amd.tsn import TSN
    net = TSN(4, base_model="resnet50", is_shift=True, shift_div=8, shift_place="block")
    full_keys = sorted(net.state_dict())
    net.base_model = torch.nn.Sequential(*list(net.base_model.children())[:-1])         # STH/evaluate.py:83
    return net, full_keys


def test_g14_shift_place_block():
    """shift_place = 'block' (STH/ops/temporal_shift.py:104-121): TemporalShift around every whole Bottleneck.  The host mirror carries
    the reference's state-dict keys for that placement (full and fc-stripped), and the oracle's trunk with the shift in front of the
    block (identity and downsample included) reproduces the reference's features."""
    g = golden("g14_sth_block_shift")
    net, full_keys = _block_shift_tsn()
    assert full_keys == g["keys_full"].tolist()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert sorted(shapes) == g["keys_stripped"].tolist()
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1414).items()}
    net.load_state_dict(sd, strict=True)                                                 # the stripped spelling loads
    csd = O.canonical_resnet_keys({"n." + k: v for k, v in sd.items()}, "n.base_model.")
    x = rnd((8, 3, 64, 64), 141)
    with torch.no_grad():
        feat = O.resnet50_trunk(csd, "n.base_model.", x, 4, 8, shift_place="block").flatten(1)
        other = O.resnet50_trunk(csd, "n.base_model.", x, 4, 8, shift_place="blockres").flatten(1)
    np.testing.assert_allclose(feat.numpy(), g["feat"], rtol=1e-4, atol=1e-4)
    assert (other - feat).abs().max().item() > 1e-2            # the two placements are different networks


def test_g15_one_step_act_validation_branch():
    """The stage-2 validation loop body, GFV.one_step_act(training=False) (ACT/models/gfv_net.py:160-210, ACT/main_dist.py:346-362), for
    both baseline kinds: the oracle's restatement against the reference's own outputs (tools/gen_golden_r6.py)."""
    g = golden("g15_one_step_act")
    sd = synth_sd("ACT", 1007)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=5))
    fr5 = frames.view(2, 8, 3, 224, 224)
    with torch.no_grad():
        fm, fv = O.glancer_act(sd, "glancer.net.", frames.view(16, 3, 224, 224))
        fm, fv = fm.view(2, 8, *fm.shape[1:]), fv.view(2, 8, -1)
        for rew in ("random", "prev"):
            state = {}
            for s in range(int(g["steps"][0])):
                org = g["random_origins"][s] if rew == "random" else None
                logits, last, psl, action, base = O.act_one_step_eval(sd, fr5[:, s], fm[:, s], fv[:, s], state, 96, reward=rew, crop_origin=org)
                assert psl is None
                assert np.array_equal(action.numpy(), g["%s_action_%d" % (rew, s)]), (rew, s)
                np.testing.assert_allclose(logits.numpy(), g["%s_logits_%d" % (rew, s)], rtol=1e-4, atol=2e-5)
                np.testing.assert_allclose(last.numpy(), g["%s_last_%d" % (rew, s)], rtol=1e-4, atol=2e-5)
                np.testing.assert_allclose(base.numpy(), g["%s_baseline_%d" % (rew, s)], rtol=1e-4, atol=2e-5)


def test_g15_stage1_form_in_eval_mode():
    """GFV.forward(one_step=False, training=False) (ACT/models/gfv_net.py:135-150) for a random_patch model (the recorded numpy crops) and for
    a policy model (one policy step over the B*T frames): the oracle's restatement against the reference's outputs."""
    g = golden("g15_one_step_act")
    sd = synth_sd("ACT", 1007)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=int(g["s1_seed_frames"][0])))
    with torch.no_grad():
        lg, last = O.act_stage1_eval(sd, frames, frames, 96, crop_origin=g["s1_random_origins"])
        np.testing.assert_allclose(lg.numpy(), g["s1_random_logits"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(last.numpy(), g["s1_random_last"], rtol=1e-4, atol=2e-5)
        lg, last = O.act_stage1_eval(sd, frames, frames, 96)
        np.testing.assert_allclose(lg.numpy(), g["s1_policy_logits"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(last.numpy(), g["s1_policy_last"], rtol=1e-4, atol=2e-5)
