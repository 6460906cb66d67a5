"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (PyTorch fp32) restatement of the AdaFocus offline-inference hot path, written
functionally over a flat state dict.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; nothing under ``adafocus_amd/`` does.

Parity pin: every function here is checked against vectors produced by importing the real
reference in the build container (``tools/gen_golden.py`` -> ``tests/golden/*.npz``,
``tests/test_oracle_golden.py``).  The arithmetic of conv/BN/GRU itself lives in PyTorch (the
reference has no native code), so "the reference's result" means torch-CPU fp32
semantics of the ops below.

Path aliases: ACT/ = "Experiments on ActivityNet, FCVID and Mini-Kinetics/",
STH/ = "Experiments on Something-Something V1&V2/" under /root/reference.
"""
import math

import re

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, used everywhere in the reference


# --------------------------------------------------------------------------------------
# a1/a2: patch gather + action tables
# --------------------------------------------------------------------------------------
def patch_coords(action_sequence, image_size, patch_size):
    """ACT/models/utils.py:42 (= STH/models/utils.py:49): fp32 multiply by the python int
    (H - P), floor, truncate to int32.  Column 0 is the row (y) origin, column 1 the x origin;
    both use image_size = H (utils.py:40)."""
    return torch.floor(action_sequence.float() * (image_size - patch_size)).int()


def get_patch(images, action_sequence, patch_size):
    """ACT/models/utils.py:37-51.  images (N,C,H,W), action (N,2) in [0,1] -> (N,C,P,P)."""
    n, c, h, _ = images.shape
    yx = patch_coords(action_sequence, h, patch_size).long()
    ar = torch.arange(patch_size)
    ys = (yx[:, 0:1] + ar)[:, None, :, None]
    xs = (yx[:, 1:2] + ar)[:, None, None, :]
    return images[torch.arange(n)[:, None, None, None], torch.arange(c)[None, :, None, None], ys, xs]


def standard_actions(action_dim):
    """ACT/models/gfv_net.py:272-307, STH/models/gfv_net.py:285-381: s x s grid,
    torch.Tensor([[i/(s-1), j/(s-1)] ...]) -- python doubles rounded to fp32."""
    s = int(round(math.sqrt(action_dim)))
    assert s * s == action_dim
    return torch.tensor([[i / (s - 1), j / (s - 1)] for i in range(s) for j in range(s)],
                        dtype=torch.float64).float()


def crop_resize(images, action_sequence, size, patch_size):
    """N1: (y, x, size) -> patch: get_patch with patch_size = size (ACT/models/utils.py:37-51), then the bilinear
    resize the reference constructs as `self.down = Resize((P, P), BILINEAR)` (ACT/models/gfv_net.py:58) -- on tensors
    torchvision's Resize is F.interpolate(mode='bilinear', align_corners=False) (torchvision 0.8
    transforms/functional_tensor.py:resize; torchvision itself is not installed here).  The reference never calls
    `self.down` on its evaluation path; with size == patch_size this is get_patch itself.
    size: int or (N,) ints (per action)."""
    n = images.shape[0]
    sizes = [int(size)] * n if not hasattr(size, "__len__") else [int(v) for v in size]
    outs = []
    for i in range(n):
        win = get_patch(images[i:i + 1], action_sequence[i:i + 1], sizes[i])
        outs.append(win if sizes[i] == patch_size else
                    F.interpolate(win, size=(patch_size, patch_size), mode="bilinear", align_corners=False))
    return torch.cat(outs, 0)


def glancer_input(images, glance_size):
    """input_prime = F.interpolate(images, (glance_size, glance_size)) -- default nearest mode
    (ACT/main_dist.py:331-332, STH/evaluate.py:188)."""
    return F.interpolate(images, (glance_size, glance_size))


# --------------------------------------------------------------------------------------
# a6: temporal shift
# --------------------------------------------------------------------------------------
def temporal_shift(x, n_segment, fold_div):
    """STH/ops/temporal_shift.py:28-46.  x (B*T,C,h,w)."""
    nt, c, h, w = x.shape
    v = x.view(nt // n_segment, n_segment, c, h, w)
    fold = c // fold_div
    out = torch.zeros_like(v)
    out[:, :-1, :fold] = v[:, 1:, :fold]
    out[:, 1:, fold:2 * fold] = v[:, :-1, fold:2 * fold]
    out[:, :, 2 * fold:] = v[:, :, 2 * fold:]
    return out.view(nt, c, h, w)


# --------------------------------------------------------------------------------------
# a4/a5: ResNet-50 trunk (optionally TSM)
# --------------------------------------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, BN_EPS)


def bottleneck(sd, p, x, stride, tsm_segments=0, tsm_div=8, shift_place="blockres"):
    """ACT/models/resnet.py:94-114; stride sits on conv2 (:86); TSM wraps conv1 (shift_place 'blockres',
    STH/ops/temporal_shift.py:123-140) or the whole block ('block', :104-121: the identity and the downsample conv see
    the shifted input too)."""
    if tsm_segments and shift_place == "block":
        x = temporal_shift(x, tsm_segments, tsm_div)
        tsm_segments = 0
    y = temporal_shift(x, tsm_segments, tsm_div) if tsm_segments else x
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(y, sd[p + ".conv1.weight"])))
    y = F.relu(_bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], stride=stride, padding=1)))
    y = _bn(sd, p + ".bn3", F.conv2d(y, sd[p + ".conv3.weight"]))
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    return F.relu(y + x)


RESNET50_STAGES = ((1, 3, 1), (2, 4, 2), (3, 6, 2), (4, 3, 2))  # (layer idx, blocks, stride)


def resnet50_stem(sd, p, x):
    """ACT/models/resnet.py:212-215."""
    y = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)))
    return F.max_pool2d(y, 3, 2, 1)


def resnet50_trunk(sd, p, x, tsm_segments=0, tsm_div=8, pooled=True, shift_place="blockres"):
    """ResNet.get_featmap(x, pooled) -- ACT/models/resnet.py:211-225.  p = key prefix such as
    'focuser.net.'.  Returns (N,2048,1,1) if pooled."""
    y = resnet50_stem(sd, p, x)
    for li, nblk, stride in RESNET50_STAGES:
        for b in range(nblk):
            y = bottleneck(sd, "%slayer%d.%d" % (p, li, b), y, stride if b == 0 else 1, tsm_segments, tsm_div, shift_place)
    return F.adaptive_avg_pool2d(y, 1) if pooled else y


# --------------------------------------------------------------------------------------
# a10: MobileNetV2 glancer (two vendored variants with different key layouts)
# --------------------------------------------------------------------------------------
MBV2_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def mbv2_blocks():
    """[(inp, oup, stride, t)] for features.1 .. features.17."""
    out, cin = [], 32
    for t, c, n, s in MBV2_SETTING:
        for i in range(n):
            out.append((cin, c, s if i == 0 else 1, t))
            cin = c
    return out


def _cbr6(sd, conv_key, bn_key, x, stride=1, pad=0, groups=1):
    return F.relu6(_bn(sd, bn_key, F.conv2d(x, sd[conv_key + ".weight"], stride=stride, padding=pad, groups=groups)))


def mobilenetv2_features(sd, p, x, variant, tsm_segments=0, tsm_div=8):
    """variant 'act': ACT/models/mobilenet.py (ConvBNReLU sub-Sequentials);
    variant 'sth': STH/models/mobilenetv2.py (flat Sequentials), optional TSM on conv[0] of the
    residual blocks (STH/models/gfv_net.py:238-241).  Returns the (N,1280,h/32,w/32) map."""
    f = p + "features."
    y = _cbr6(sd, f + "0.0", f + "0.1", x, 2, 1)
    for i, (inp, oup, stride, t) in enumerate(mbv2_blocks(), start=1):
        b = "%s%d.conv." % (f, i)
        hid = inp * t
        z = y
        if variant == "act":
            k = 0
            if t != 1:
                z = _cbr6(sd, b + "0.0", b + "0.1", z)
                k = 1
            z = _cbr6(sd, b + "%d.0" % k, b + "%d.1" % k, z, stride, 1, hid)
            z = _bn(sd, b + "%d" % (k + 2), F.conv2d(z, sd[b + "%d.weight" % (k + 1)]))
        else:
            if t != 1:
                res = stride == 1 and inp == oup
                if tsm_segments and res:
                    z = temporal_shift(z, tsm_segments, tsm_div)
                    z = F.relu6(_bn(sd, b + "1", F.conv2d(z, sd[b + "0.net.weight"])))
                else:
                    z = _cbr6(sd, b + "0", b + "1", z)
                z = _cbr6(sd, b + "3", b + "4", z, stride, 1, hid)
                z = _bn(sd, b + "7", F.conv2d(z, sd[b + "6.weight"]))
            else:
                z = _cbr6(sd, b + "0", b + "1", z, stride, 1, hid)
                z = _bn(sd, b + "4", F.conv2d(z, sd[b + "3.weight"]))
        y = y + z if (stride == 1 and inp == oup) else z
    return _cbr6(sd, f + "18.0", f + "18.1", y)


def glancer_act(sd, p, x):
    """Glancer.forward -> MobileNetV2.get_featmap: ACT/models/mobilenet.py:146-148."""
    fm = mobilenetv2_features(sd, p, x, "act")
    return fm, fm.mean([2, 3])


def glancer_sth(sd, p, x, tsm_segments, tsm_div):
    """STH/models/mobilenetv2.py:116-121: (featmap, classifier(mean(3).mean(2)))."""
    fm = mobilenetv2_features(sd, p, x, "sth", tsm_segments, tsm_div)
    return fm, F.linear(fm.mean(3).mean(2), sd[p + "classifier.weight"], sd[p + "classifier.bias"])


# --------------------------------------------------------------------------------------
# GRU cell (PyTorch gate order r,z,n) -- used by the policy and by the classifier
# --------------------------------------------------------------------------------------
def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def _gru_params(sd, p):
    return sd[p + "weight_ih_l0"], sd[p + "weight_hh_l0"], sd[p + "bias_ih_l0"], sd[p + "bias_hh_l0"]


# --------------------------------------------------------------------------------------
# a11: policy (eval branch only)
# --------------------------------------------------------------------------------------
def policy_encode_act(sd, p, state):
    """ACT/models/ppo.py:33-39: conv1x1(no bias) -> ReLU -> flatten -> Linear -> ReLU."""
    y = F.relu(F.conv2d(state, sd[p + "state_encoder.0.weight"]))
    return F.relu(F.linear(y.flatten(1), sd[p + "state_encoder.3.weight"], sd[p + "state_encoder.3.bias"]))


def policy_act_discrete(sd, p, state, hidden, return_gap=False):
    """One step of ActorCritic.act(training=False) -- ACT/models/ppo.py:67-96.  Returns
    (action index (B,) int64, new hidden (B,Hd)); with return_gap also log p(top-1) - log p(top-2) per clip, the margin
    an arg-max flip would have to overcome (tests assert it before comparing policy-driven outputs)."""
    h = gru_cell(policy_encode_act(sd, p, state), hidden, *_gru_params(sd, p + "gru."))
    probs = torch.softmax(F.linear(h, sd[p + "actor.0.weight"], sd[p + "actor.0.bias"]), dim=-1)
    if return_gap:
        top = torch.log(probs).topk(2, dim=1)[0]
        return probs.max(1)[1], h, top[:, 0] - top[:, 1]
    return probs.max(1)[1], h


def policy_act_continuous(sd, p, state, hidden, with_bn=True):
    """STH/models/ppo_continuous.py:78-109 eval branch: action = action_mean = sigmoid(actor(h))."""
    e = p + "state_encoder."
    y = F.conv2d(state, sd[e + "0.weight"])
    if with_bn:
        y = F.relu(_bn(sd, e + "1", y))
        y = F.linear(y.flatten(1), sd[e + "4.weight"], sd[e + "4.bias"])
        y = F.relu(F.batch_norm(y, sd[e + "5.running_mean"], sd[e + "5.running_var"], sd[e + "5.weight"],
                                sd[e + "5.bias"], False, 0.0, BN_EPS))
    else:
        y = F.relu(F.linear(F.relu(y).flatten(1), sd[e + "3.weight"], sd[e + "3.bias"]))
    h = gru_cell(y, hidden, *_gru_params(sd, p + "gru."))
    return torch.sigmoid(F.linear(h, sd[p + "actor.0.weight"], sd[p + "actor.0.bias"])), h


# --------------------------------------------------------------------------------------
# a7/a8: aggregation
# --------------------------------------------------------------------------------------
def recurrent_classifier(sd, p, feature):
    """RecurrentClassifier.forward -- ACT/models/gfv_net.py:427-435.  feature (B,T,F) ->
    (logits (B*T,C), last (B,C)); h0 = 0, dropout is identity in eval."""
    b, t, _ = feature.shape
    w = _gru_params(sd, p + "gru.")
    h = feature.new_zeros(b, w[1].shape[1])
    outs = []
    for s in range(t):
        h = gru_cell(feature[:, s], h, *w)
        outs.append(h)
    out = torch.stack(outs, 1)
    logits = F.linear(out.reshape(b * t, -1), sd[p + "fc.weight"], sd[p + "fc.bias"])
    return logits, logits.view(b, t, -1)[:, -1, :].reshape(b, -1)


def fc_consensus(sd, p, feat, batch, global_logit=None):
    """STH/models/gfv_net.py:164-174: mean_t FC(f_t) (+ mean_t glancer logits).
    feat (B*T,2048), global_logit (B,Tg,C)."""
    logit = F.linear(feat, sd[p + "weight"], sd[p + "bias"]).view(batch, -1, sd[p + "weight"].shape[0])
    out = logit.mean(dim=1, keepdim=True).squeeze(1)
    if global_logit is not None:
        out = global_logit.mean(dim=1, keepdim=True).squeeze(1) + out
    return out


def linear_classifier(sd, p, feature):
    """LinearCLassifier.forward -- ACT/models/gfv_net.py:399-407: softmax of FC per step, mean over steps;
    returns (log(avg), avg).  feature (B,T,F)."""
    b, t, _ = feature.shape
    logits = F.linear(feature.reshape(b * t, -1), sd[p + "fc.weight"], sd[p + "fc.bias"])
    avg = torch.softmax(logits, dim=1).view(b, t, -1).mean(dim=1)
    return torch.log(avg), avg


def backbone_pred(sd, images, which):
    """GFV.forward(backbone_pred=True) -- ACT/models/gfv_net.py:85-94: per-frame class logits of the glancer
    (MobileNetV2 + its classifier) or the focuser (ResNet-50 + fc) on FULL frames.  images (B,T*3,H,W) -> (B,T,C)."""
    b, tc, hh, ww = images.shape
    x = images.view(b * (tc // 3), 3, hh, ww)
    if which == "glancer":
        fm = mobilenetv2_features(sd, "glancer.net.", x, "act")
        out = F.linear(fm.mean([2, 3]), sd["glancer.net.classifier.1.weight"], sd["glancer.net.classifier.1.bias"])
    else:
        f = resnet50_trunk(sd, "focuser.net.", x).flatten(1)
        out = F.linear(f, sd["focuser.net.fc.weight"], sd["focuser.net.fc.bias"])
    return out.view(b, tc // 3, -1)


# --------------------------------------------------------------------------------------
# a9: end-to-end compositions
# --------------------------------------------------------------------------------------
def act_forward(sd, images, scan, patch_size, action_dim=49, forced_action_idx=None, per_step=True,
                return_aux=False, return_gap=False):
    """GFV.forward(one_step=True, training=False) -- ACT/models/gfv_net.py:95-133.

    per_step=True follows the reference's structure literally (T sequential focuser calls of
    batch B); per_step=False is the offline restructuring (all actions, one batched crop, one
    batched local-CNN pass, one GRU scan) that SURVEY.md §0.4 shows is exactly equivalent.
    forced_action_idx (B,T) int64 overrides the policy's argmax (parity tests need varied crops).
    return_gap appends the policy's arg-max margins (B,T) (policy_act_discrete) to the result.
    """
    b, tc, hh, ww = images.shape
    t = tc // 3
    frames = images.view(b, t, 3, hh, ww)
    fm, fv = glancer_act(sd, "glancer.net.", scan.reshape(b * t, 3, scan.shape[2], scan.shape[3]))
    fm = fm.view(b, t, *fm.shape[1:])
    fv = fv.view(b, t, -1)
    table = standard_actions(action_dim)
    pol = "focuser.policy.policy_old."
    hid = images.new_zeros(b, sd[pol + "gru.weight_hh_l0"].shape[1])
    feats, idx_all, gaps = [], [], []
    if per_step:
        for s in range(t):
            idx, hid, gap = policy_act_discrete(sd, pol, fm[:, s], hid, return_gap=True)
            gaps.append(gap)
            if forced_action_idx is not None:
                idx = forced_action_idx[:, s]
            idx_all.append(idx)
            patch = get_patch(frames[:, s], table[idx], patch_size)
            local = resnet50_trunk(sd, "focuser.net.", patch).view(b, -1)
            feats.append(torch.cat([fv[:, s], local], dim=1))
        feature = torch.stack(feats, dim=1)
        idx_all = torch.stack(idx_all, 1)
    else:
        for s in range(t):
            idx, hid, gap = policy_act_discrete(sd, pol, fm[:, s], hid, return_gap=True)
            gaps.append(gap)
            idx_all.append(idx)
        idx_all = torch.stack(idx_all, 1) if forced_action_idx is None else forced_action_idx
        patch = get_patch(frames.reshape(b * t, 3, hh, ww), table[idx_all.reshape(-1)], patch_size)
        local = resnet50_trunk(sd, "focuser.net.", patch).view(b, t, -1)
        feature = torch.cat([fv, local], dim=2)
    out = recurrent_classifier(sd, "classifier.", feature)
    if return_aux:
        out = out + (idx_all, feature)
    return out + (torch.stack(gaps, 1),) if return_gap else out


def act_one_step_eval(sd, img, glancer_map, glancer_vec, state, patch_size, action_dim=49, reward="random", crop_origin=None):
    """GFV.one_step_act(img, map, vec, restart_batch, training=False) -- ACT/models/gfv_net.py:160-210, the body of the stage-2 VALIDATION
    loop (ACT/main_dist.py:346-362), with the classifier's step functions (gfv_net.py:437-457): `test_single_forward` runs the baseline
    feature through the GRU from the CURRENT hidden state without storing the result, `single_forward` advances it.
    img (B,3,H,W), glancer_map (B,1280,h,w), glancer_vec (B,1280); `state` = {} at restart_batch (the zero states of ppo.py:69-70 and
    gfv_net.py:439-440) and is updated in place.  reward: 'random' -> the baseline's local feature comes from one crop per clip at
    `crop_origin` (B,2) integer (y, x) (utils.py:24-35 draws them with np.random.randint); 'padding' | 'prev' | 'conf' -> zeros.
    Returns (logits (B,C), last_out (B,C), None, standard action (B,2), baseline logits (B,C))."""
    b = img.shape[0]
    pol = "focuser.policy.policy_old."
    if "policy_h" not in state:
        state["policy_h"] = img.new_zeros(b, sd[pol + "gru.weight_hh_l0"].shape[1])
        state["hx"] = img.new_zeros(b, sd["classifier.gru.weight_hh_l0"].shape[1])
    idx, state["policy_h"] = policy_act_discrete(sd, pol, glancer_map, state["policy_h"])
    action = standard_actions(action_dim)[idx]
    local = resnet50_trunk(sd, "focuser.net.", get_patch(img, action, patch_size)).view(b, -1)
    if reward == "random":
        crops = torch.stack([img[i, :, int(y):int(y) + patch_size, int(x):int(x) + patch_size] for i, (y, x) in enumerate(crop_origin)])
        base_local = resnet50_trunk(sd, "focuser.net.", crops).view(b, -1)
    elif reward in ("padding", "prev", "conf"):
        base_local = torch.zeros_like(local)
    else:
        raise NotImplementedError(reward)
    w = _gru_params(sd, "classifier.gru.")
    fcw, fcb = sd["classifier.fc.weight"], sd["classifier.fc.bias"]
    base_logits = F.linear(gru_cell(torch.cat([glancer_vec, base_local], 1), state["hx"], *w), fcw, fcb)
    state["hx"] = gru_cell(torch.cat([glancer_vec, local], 1), state["hx"], *w)
    logits = F.linear(state["hx"], fcw, fcb)
    return logits, logits, None, action, base_logits


def act_stage1_eval(sd, images, scan, patch_size, action_dim=49, crop_origin=None):
    """GFV.forward(one_step=False, training=False) -- ACT/models/gfv_net.py:135-150, the stage-1 form validate() runs at train_stage 1
    (ACT/main_dist.py:334-340): glancer over all B*T frames, focuser over all B*T frames AT ONCE, classifier.
    crop_origin (B*T,2) integer (y, x): a random_patch model (gfv_net.py:317-327 -> utils.py:24-35, one random crop per frame);
    None: a policy model -- ONE ActorCritic.act step over the B*T frames as a batch from the zero hidden state."""
    b, tc, hh, ww = images.shape
    t = tc // 3
    frames = images.view(b * t, 3, hh, ww)
    fm, fv = glancer_act(sd, "glancer.net.", scan.reshape(b * t, 3, scan.shape[2], scan.shape[3]))
    if crop_origin is not None:
        patch = torch.stack([frames[i, :, int(y):int(y) + patch_size, int(x):int(x) + patch_size] for i, (y, x) in enumerate(crop_origin)])
    else:
        pol = "focuser.policy.policy_old."
        idx, _ = policy_act_discrete(sd, pol, fm, images.new_zeros(b * t, sd[pol + "gru.weight_hh_l0"].shape[1]))
        patch = get_patch(frames, standard_actions(action_dim)[idx], patch_size)
    local = resnet50_trunk(sd, "focuser.net.", patch).view(b * t, -1)
    return recurrent_classifier(sd, "classifier.", torch.cat([fv, local], dim=1).view(b, t, -1))


def act_hot_path(sd, frames_nchw, glancer_vec, actions, patch_size):
    """The benchmarked slice of act_forward: batched crop -> local CNN -> concat -> GRU+FC.
    frames (B*T,3,H,W); glancer_vec (B,T,1280); actions (B*T,2)."""
    b, t, _ = glancer_vec.shape
    patch = get_patch(frames_nchw, actions, patch_size)
    local = resnet50_trunk(sd, "focuser.net.", patch).view(b, t, -1)
    return recurrent_classifier(sd, "classifier.", torch.cat([glancer_vec, local], dim=2))


def sth_forward(sd, glancer_images, focuser_images, patch_size, tg, tf, shift_div=8, forced_action=None,
                net_prefix="focuser.net.base_model."):
    """STH/evaluate.py:195-201 with video_div=1, main (non-baseline) branch of
    GFV.action_stage2 -- STH/models/gfv_net.py:136-174.  glancer_images (B,Tg*3,H,W),
    focuser_images (B,Tf,3,H,W).  Returns (total_logit (B,C), local_patch (B,Tf,3,P,P), action)."""
    b = glancer_images.shape[0]
    hh, ww = glancer_images.shape[2:]
    fm, gl = glancer_sth(sd, "glancer.net.", glancer_images.view(b * tg, 3, hh, ww), tg, shift_div)
    fm = fm.view(b, tg, *fm.shape[1:])
    gl = gl.view(b, tg, -1)
    pol = "policy."
    state = fm.view(b, -1, fm.shape[3], fm.shape[4])
    hid = fm.new_zeros(b, sd[pol + "gru.weight_hh_l0"].shape[1])
    action, _ = policy_act_continuous(sd, pol, state, hid, with_bn=(pol + "state_encoder.1.running_mean") in sd)
    if forced_action is not None:
        action = forced_action
    cur = focuser_images.view(b, -1, focuser_images.shape[3], focuser_images.shape[4])
    patch = get_patch(cur, action, patch_size).view(b, tf, 3, patch_size, patch_size)
    feat = resnet50_trunk(sd, net_prefix, patch.view(-1, 3, patch_size, patch_size), tf, shift_div).squeeze()
    return fc_consensus(sd, "classifier.", feat, b, gl), patch, action


def sth_stage(sd, glancer_map, glancer_logit, focuser_images, step, video_div, patch_size, tf, hidden, prev_local_patch=None,
              shift_div=8, forced_action=None, baseline_action=None, net_prefix="focuser.net.base_model."):
    """One call of GFV.action_stage2(training=False) -- STH/models/gfv_net.py:136-188 -- for any video_div: the policy sees
    the glancer maps of segment `step` concatenated on the channel axis and carries its GRU state (`hidden`, zeros when
    step == 0: restart_batch), the new patches are appended to the previous steps' (`prev_local_patch`), and the TSM trunk
    runs over ALL patches so far with its construction-time n_segment = tf (tsn.py / temporal_shift.py:103).  The reward
    baseline (random_patching, :153-160,176-186) is evaluated when `baseline_action` (the torch.rand draw) is given.
    glancer_map (B,Tg,1280,h,w); glancer_logit (B,Tg,C); focuser_images (B,Tf,3,H,W).
    Returns (total_logit, baseline_logit or None, local_patch, action, hidden)."""
    b = focuser_images.shape[0]
    nfg, nff = glancer_map.shape[1] // video_div, tf // video_div
    cur = focuser_images[:, step * nff:(step + 1) * nff].reshape(b, -1, focuser_images.shape[3], focuser_images.shape[4])
    state = glancer_map[:, step * nfg:(step + 1) * nfg].reshape(b, -1, glancer_map.shape[3], glancer_map.shape[4])
    pol = "policy."
    if hidden is None:
        hidden = state.new_zeros(b, sd[pol + "gru.weight_hh_l0"].shape[1])
    action, hidden = policy_act_continuous(sd, pol, state, hidden, with_bn=(pol + "state_encoder.1.running_mean") in sd)
    if forced_action is not None:
        action = forced_action

    def branch(act):
        cur_patch = get_patch(cur, act, patch_size).view(b, nff, 3, patch_size, patch_size)
        patch = cur_patch if prev_local_patch is None else torch.cat([prev_local_patch, cur_patch], dim=1)
        feat = resnet50_trunk(sd, net_prefix, patch.reshape(-1, 3, patch_size, patch_size), tf, shift_div).flatten(1)
        return fc_consensus(sd, "classifier.", feat, b, glancer_logit), patch

    total, local_patch = branch(action)
    base = branch(baseline_action)[0] if baseline_action is not None else None
    return total, base, local_patch, action, hidden


# --------------------------------------------------------------------------------------
# state-dict key helpers (the STH checkpoints carry wrapper / Sequential-index names)
# --------------------------------------------------------------------------------------
_SEQ_TO_NAME = {"0": "conv1", "1": "bn1", "4": "layer1", "5": "layer2", "6": "layer3", "7": "layer4"}


def canonical_resnet_keys(sd, prefix):
    """Map the STH focuser's keys to torchvision names under the same prefix:
    'base_model.4.0.conv1.net.weight' (after STH/evaluate.py:83 strips fc by re-wrapping the
    children in a Sequential, and STH/ops/temporal_shift.py:123-140 wraps conv1) ->
    'base_model.layer1.0.conv1.weight'.  Other keys pass through."""
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            rest = k[len(prefix):].replace(".conv1.net.", ".conv1.")
            rest = re.sub(r"^((?:layer)?\d\.\d+)\.net\.", r"\1.", rest)      # shift_place = 'block': TemporalShift(Bottleneck), :104-121
            head, _, tail = rest.partition(".")
            if head in _SEQ_TO_NAME:
                rest = _SEQ_TO_NAME[head] + "." + tail
            out[prefix + rest] = v
        else:
            out[k] = v
    return out


# --------------------------------------------------------------------------------------
# f1: frame ingest (loader tail): Stack -> ToTorchFormatTensor -> GroupNormalize
# --------------------------------------------------------------------------------------
INPUT_MEAN = (0.485, 0.456, 0.406)   # GFV.input_mean / input_std, ACT/models/gfv_net.py:29-30
INPUT_STD = (0.229, 0.224, 0.225)


def ingest_uint8(stacked_hwc_u8, mean=INPUT_MEAN, std=INPUT_STD):
    """stacked (H, W, T*3) uint8 numpy array of one clip -> (T*3, H, W) fp32, the reference's exact op
    sequence: torch.from_numpy(pic).permute(2,0,1).contiguous().float().div(255)
    (ACT/ops/transforms.py:325-336) then per channel t.sub_(m).div_(s) with the mean/std lists
    repeated T times (:69-77)."""
    img = torch.from_numpy(stacked_hwc_u8).permute(2, 0, 1).contiguous().float().div(255)
    reps = img.size(0) // len(mean)
    for t, m, s in zip(img, list(mean) * reps, list(std) * reps):
        t.sub_(m).div_(s)
    return img
