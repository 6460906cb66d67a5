"""Something-Something V1/V2 model composition -- host-side mirror of STH/models/gfv_net.py for
offline inference (the call sequence of STH/evaluate.py:195-213 and stage-3 validation).

Differences from the ActivityNet model: one (y,x) per CLIP from a continuous policy applied to all
T focuser frames (``get_patch`` on a (B, 3T, H, W) view, STH/models/gfv_net.py:148-152), TSM-ResNet-50
local CNN (shift fused into conv1), and a linear classifier with temporal-mean consensus added to the
glancer's mean logits (:164-174).  Hot path here: ONE gather launch with ``frames_per_action = T``
(NHWC4), ONE TSM trunk pass over B*T patches -- the reward-baseline branch (:153-160,176-186), when
requested, rides in the SAME pass as a second half of the batch -- then ``adaf_fc_meanpool_forward_f32``.
"""
import math

import torch
from torch import nn

from . import hip_ops
from .basic_ops import ConsensusModule
from .mobilenetv2 import InvertedResidual, mobilenet_v2
from .ppo import PPO, Memory
from .ppo_continuous import PPO_Continuous
from .synth import grid_table
from .temporal_shift import TemporalShift
from .tsn import TSN
from .utils import get_patch, get_patch_nhwc4, nchw_to_nhwc4

__all__ = ["GFV", "Glancer", "Focuser", "PatchSampler"]


class _FcMeanPool(torch.autograd.Function):
    """mean_t FC(f_t) (+ mean_t glancer logits) with a backward for the CLASSIFIER parameters -- all that stage 3 of this
    build trains (STH/stage3.py:351-353 with the local CNN frozen).  Forward = adaf_fc_meanpool_forward_f32; backward =
    two GEMMs on the conv engine:  dW = g^T . mean_t(f)  and  db = g^T . 1  (d mean_t FC(f_t) / dW = mean_t f_t)."""

    @staticmethod
    def forward(ctx, feat, weight, bias, glog, batch):
        ctx.save_for_backward(feat)
        ctx.batch = batch
        return hip_ops.fc_meanpool_forward(feat, batch, weight.detach(), bias.detach(), glog)

    @staticmethod
    def backward(ctx, g):
        (feat,) = ctx.saved_tensors
        b = ctx.batch
        t = feat.shape[0] // b
        mean_feat = hip_ops.global_avgpool(feat.view(b, t, 1, -1))              # (B, F) = mean_t f_t
        bp = (b + 3) // 4 * 4                                                     # the engine wants K % 4 == 0
        gt = torch.zeros((g.shape[1], bp), device=g.device, dtype=torch.float32)
        gt[:, :b] = g.t()
        mt = torch.zeros((mean_feat.shape[1], bp), device=g.device, dtype=torch.float32)
        mt[:, :b] = mean_feat.t()
        ones = torch.zeros((4, bp), device=g.device, dtype=torch.float32)
        ones[0, :b] = 1.0
        dw = hip_ops.linear(gt, mt)                                               # (C, F)
        db = hip_ops.linear(gt, ones)[:, 0].contiguous()                          # (C,)
        return None, dw, db, None, None


class GFV(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.num_segments_glancer = args.num_segments_glancer
        self.num_segments_focuser = args.num_segments_focuser
        self.num_class = args.num_classes
        self.input_size = 224
        self.batch_size = args.batch_size
        self.patch_size = args.patch_size
        self.input_mean = [0.485, 0.456, 0.406]
        self.input_std = [0.229, 0.224, 0.225]
        self.with_glancer = args.with_glancer
        self.glancer = Glancer(args)
        tg = args.num_segments_glancer // args.video_div
        cells = math.ceil(args.glance_size / 32)
        policy_params = dict(feature_dim=args.feature_map_channels * tg,
                             state_dim=args.feature_map_channels * tg * cells * cells, action_dim=args.action_dim,
                             hidden_state_dim=args.hidden_state_dim, policy_conv=args.policy_conv, gpu=args.gpu,
                             frame_channels=args.feature_map_channels,
                             ppo_continuous=args.ppo_continuous, gamma=args.gamma, policy_lr=args.policy_lr,
                             action_std=args.action_std, with_bn=args.actorcritic_with_bn)
        base = dict(num_segments=args.num_segments_focuser, modality=args.modality, base_model=args.base_model,
                    partial_bn=args.partial_bn, pretrain=args.pretrain, is_shift=args.is_shift, shift_div=args.shift_div,
                    shift_place=args.shift_place, fc_lr5=args.fc_lr5, temporal_pool=args.temporal_pool,
                    non_local=args.non_local)
        self.focuser = Focuser(args.patch_size, args.random_patch, policy_params, base)
        self.dropout = nn.Dropout(p=args.dropout)
        self.classifier = nn.Linear(in_features=self.focuser.feature_dim, out_features=args.num_classes)
        self.consensus = ConsensusModule(consensus_type="avg")

    def _apply(self, fn, *a, **k):   # the continuous policy is a plain holder, move it with the model
        super()._apply(fn, *a, **k)
        pol = self.focuser.policy
        if pol is not None and not isinstance(pol, nn.Module):
            pol.policy._apply(fn)
            pol.policy_old._apply(fn)
        return self

    # ---- reference surface --------------------------------------------------------------
    def glance(self, input_prime):
        """Reference layout: featmap (B,T,1280,h,w) as a permuted view of the pixel-major map, logits (B,T,C)."""
        b, tc, hh, ww = input_prime.shape
        t = tc // 3
        fm, logit = self.glancer(input_prime.reshape(b * t, 3, hh, ww))
        return fm.unflatten(0, (b, t)), logit.view(b, t, -1)

    def _stage(self, focuser_image, global_feat_map, global_feat_logit, step, args, prev_local_patch, with_baseline,
               forced_action=None, baseline_action=None, train_classifier=False):
        if self.training and not train_classifier:
            raise RuntimeError("adafocus_amd: eval mode only (stage 3 trains the classifier through action_stage3)")
        with torch.no_grad():
            feat, local_patch, action, b, frames_total, ngroups = self._stage_features(
                focuser_image, global_feat_map, step, args, prev_local_patch, with_baseline, forced_action, baseline_action)
        per = b * frames_total
        glog = global_feat_logit if self.with_glancer else None
        if train_classifier:
            # stage 3 (STH/stage3.py:351-353): the patches and local features are constants, the classifier learns
            f = self.dropout(feat[:per]) if self.training else feat[:per]
            return [_FcMeanPool.apply(f, self.classifier.weight, self.classifier.bias, glog, b)], local_patch, action
        with torch.no_grad():
            logits = [hip_ops.fc_meanpool_forward(feat[g * per:(g + 1) * per], b, self.classifier.weight.detach(),
                                                  self.classifier.bias.detach(), glog) for g in range(ngroups)]
        return logits, local_patch, action

    def _stage_features(self, focuser_image, global_feat_map, step, args, prev_local_patch, with_baseline, forced_action,
                        baseline_action):
        nfg = args.num_segments_glancer // args.video_div
        nff = args.num_segments_focuser // args.video_div
        b, _, c, hh, ww = focuser_image.shape
        cur = focuser_image[:, step * nff:(step + 1) * nff].reshape(b * nff, c, hh, ww)
        fb, _, fc_, fh, fw = global_feat_map.shape
        seg = global_feat_map[:, step * nfg:(step + 1) * nfg]
        if self.focuser.ppo_continuous:
            # pixel-major view of the map (free when it came from glance()) -> policy on the engine; the GRU state of
            # step i-1 is carried in focuser.memory.hidden for video_div > 1 (STH/evaluate.py:198, ppo_continuous.py:81-94)
            nhwc = seg.permute(0, 1, 3, 4, 2).contiguous().view(fb * nfg, fh, fw, fc_)
            action = self.focuser.policy.policy_old.act_nhwc(nhwc, fb, nfg, self.focuser.memory, restart_batch=(step == 0))
        else:
            action = self.focuser.act(seg.reshape(fb, -1, fh, fw), restart_batch=(step == 0))
        if forced_action is not None:
            action = forced_action.to(action.device)
        p = args.patch_size
        cur_patch = get_patch(cur.view(b, nff * c, hh, ww), action, p).view(b, nff, 3, p, p)     # API-layout return value
        local_patch = cur_patch if prev_local_patch is None else torch.cat([prev_local_patch, cur_patch], dim=1)
        frames_total = local_patch.shape[1]
        if prev_local_patch is None:
            main4 = get_patch_nhwc4(cur, action, p, nff)             # gather straight into the trunk's layout
        else:
            main4 = nchw_to_nhwc4(local_patch.reshape(b * frames_total, 3, p, p))
        groups = [main4]
        if with_baseline:
            # Focuser.random_patching, gfv_net.py:424-427 (`baseline_action` injects the draw for parity tests)
            rand_action = baseline_action.to(cur.device) if baseline_action is not None else torch.rand(b, 2).to(cur.device)
            base_cur = get_patch(cur.view(b, nff * c, hh, ww), rand_action, p).view(b, nff, 3, p, p)
            base_patch = base_cur if prev_local_patch is None else torch.cat([prev_local_patch, base_cur], dim=1)
            groups.append(nchw_to_nhwc4(base_patch.reshape(b * frames_total, 3, p, p)))
        # the temporal shift views its input as clips of num_segments_focuser frames -- fixed at construction, as in the
        # reference (tsn.py / temporal_shift.py:103): with video_div > 1 the partial passes shift across clip pairs
        feat = self.focuser.net.features_nhwc4(torch.cat(groups, 0) if len(groups) > 1 else groups[0])
        return feat, local_patch, action, b, frames_total, len(groups)

    # ---- the same two calls for frames that are already normalised pixel-major (uint8 ingest, evaluate.validate_sth) ----------
    @torch.no_grad()
    def glance_nhwc4(self, frames_nhwc4, b):
        """`glance` for (B*Tg, g, g, 4) pixel-major frames (transforms.ingest_uint8): -> (featmap (B*Tg, h, w, 1280) pixel-major,
        logits (B, Tg, C)) -- STH/models/gfv_net.py:101-107 without the layout round trip."""
        net = self.glancer.net
        fm4, fvec = net._engine.features(frames_nhwc4, net.tsm_segments, net.tsm_div)
        logit = hip_ops.linear(fvec, net.classifier.weight.detach(), net.classifier.bias.detach())
        return fm4, logit.view(b, -1, logit.shape[-1])

    @torch.no_grad()
    def action_stage2_nhwc4(self, focuser_frames4, featmap_nhwc, global_feat_logit, step, args, prev_patch4=None, with_baseline=True,
                            forced_action=None, baseline_action=None):
        """`action_stage2(training=False)` (STH/models/gfv_net.py:136-188) on pixel-major data end to end: focuser_frames4
        (B*Tf, H, W, 4), featmap_nhwc (B*Tg, h, w, 1280) from glance_nhwc4; the patches stay pixel-major ((B, frames, P, P, 4),
        carried from step to step as `prev_patch4`).  -> (total_logit, baseline_logit | None, patches4).  Same kernels and the
        same values as action_stage2 on the corresponding planar tensors."""
        if self.training:
            raise RuntimeError("adafocus_amd: eval mode only")
        nfg = args.num_segments_glancer // args.video_div
        nff = args.num_segments_focuser // args.video_div
        tf_ = args.num_segments_focuser
        n, hh, ww, _ = focuser_frames4.shape
        b = n // tf_
        cur4 = focuser_frames4.view(b, tf_, hh, ww, 4)[:, step * nff:(step + 1) * nff].reshape(b * nff, hh, ww, 4)
        _, fh, fw, fc_ = featmap_nhwc.shape
        seg = featmap_nhwc.view(b, args.num_segments_glancer, fh, fw, fc_)[:, step * nfg:(step + 1) * nfg].reshape(b * nfg, fh, fw, fc_)
        if self.focuser.ppo_continuous:
            action = self.focuser.policy.policy_old.act_nhwc(seg, b, nfg, self.focuser.memory, restart_batch=(step == 0))
        else:
            state = seg.view(b, nfg, fh, fw, fc_).permute(0, 1, 4, 2, 3).reshape(b, nfg * fc_, fh, fw)
            action = self.focuser.act(state, restart_batch=(step == 0))
        if forced_action is not None:
            action = forced_action.to(action.device)
        p = args.patch_size
        main4 = hip_ops.crop_gather_nhwc4(cur4, action.to(dtype=torch.float32), p, nff).view(b, nff, p, p, 4)
        patches4 = main4 if prev_patch4 is None else torch.cat([prev_patch4, main4], dim=1)
        frames_total = patches4.shape[1]
        groups = [patches4.reshape(b * frames_total, p, p, 4)]
        if with_baseline:
            # Focuser.random_patching, gfv_net.py:424-427: the reference draws on the CPU generator
            rand_action = baseline_action if baseline_action is not None else torch.rand(b, 2)
            rand_action = rand_action.to(cur4.device, non_blocking=True)
            base4 = hip_ops.crop_gather_nhwc4(cur4, rand_action, p, nff).view(b, nff, p, p, 4)
            base_all = base4 if prev_patch4 is None else torch.cat([prev_patch4, base4], dim=1)
            groups.append(base_all.reshape(b * frames_total, p, p, 4))
        feat = self.focuser.net.features_nhwc4(torch.cat(groups, 0) if len(groups) > 1 else groups[0])
        per = b * frames_total
        glog = global_feat_logit if self.with_glancer else None
        logits = [hip_ops.fc_meanpool_forward(feat[k * per:(k + 1) * per], b, self.classifier.weight.detach(),
                                              self.classifier.bias.detach(), glog) for k in range(len(groups))]
        return logits[0], (logits[1] if with_baseline else None), patches4

    def action_stage2(self, focuser_image, global_feat_map, global_feat_logit, focus_time_step, args,
                      prev_local_patch=None, training=True, with_baseline=True, forced_action=None, baseline_action=None):
        """STH/models/gfv_net.py:136-188 -> (total_logit, baseline_logit, local_patch)."""
        if training:
            raise NotImplementedError("stage-2 policy training is out of scope; call with training=False")
        logits, local_patch, _ = self._stage(focuser_image, global_feat_map, global_feat_logit, focus_time_step, args,
                                             prev_local_patch, with_baseline, forced_action, baseline_action)
        return logits[0], (logits[1] if with_baseline else None), local_patch

    def action_stage3(self, focuser_image, global_feat_map, global_feat_logit, focus_time_step, args,
                      prev_local_patch=None, forced_action=None):
        """STH/models/gfv_net.py:190-225 -> (total_logit, local_patch).  In eval mode this is the inference forward.  With
        grad enabled and the classifier's parameters requiring grad (stage-3 training, STH/stage3.py:313-317,351-353:
        model.train() with glancer / focuser / policy in eval mode) `total_logit` carries an autograd graph into
        classifier.weight / classifier.bias; the local CNN runs without grad on the HIP trunk (its fine-tuning -- the
        reference's optimizer also lists the focuser backbone -- is training code beyond this build's scope)."""
        train = self.training and torch.is_grad_enabled() and self.classifier.weight.requires_grad
        logits, local_patch, _ = self._stage(focuser_image, global_feat_map, global_feat_logit, focus_time_step, args,
                                             prev_local_patch, False, forced_action, train_classifier=train)
        return logits[0], local_patch

    @torch.no_grad()
    def forward(self, *argv, **kwargs):
        """STH/models/gfv_net.py:74-99 (eval): glance, one clip-level action, local CNN, consensus sum."""
        if kwargs.get("training"):
            raise NotImplementedError("training forward is out of scope")
        focuser_input, glancer_input = kwargs["input"], kwargs["scan"]
        b, tc, hh, ww = focuser_input.shape
        fm, glog = self.glance(glancer_input)

        class _A:
            pass
        a = _A()
        a.num_segments_glancer, a.num_segments_focuser, a.video_div, a.patch_size = \
            self.num_segments_glancer, self.num_segments_focuser, 1, self.patch_size
        logits, _, _ = self._stage(focuser_input.view(b, tc // 3, 3, hh, ww), fm, glog, 0, a, None, False)
        return logits[0]

    def adjust_patch_size(self, patch_size):
        self.patch_size = patch_size
        self.focuser.patch_size = patch_size
        self.focuser.patch_sampler.size = patch_size

    @property
    def scale_size(self):
        return self.input_size * 256 // 224

    @property
    def crop_size(self):
        return self.input_size


class Glancer(nn.Module):
    """TSM-MobileNetV2 glancer (STH/models/gfv_net.py:228-250) on adaf_mobilenetv2; the TemporalShift
    wrappers exist for state-dict key compatibility, the shift itself is fused into the expand conv."""

    def __init__(self, args, skip=False):
        super().__init__()
        self.net = mobilenet_v2(n_class=args.num_classes, pretrained=False)
        for m in self.net.modules():
            if isinstance(m, InvertedResidual) and len(m.conv) == 8 and m.use_res_connect:
                m.conv[0] = TemporalShift(m.conv[0], n_segment=args.num_segments_glancer, n_div=args.shift_div)
        self.net.tsm_segments, self.net.tsm_div = args.num_segments_glancer, args.shift_div
        self.skip = skip

    def forward(self, input):
        return self.net.get_featmap(input)

    def predict(self, input):
        return self.net(input)

    @property
    def feature_dim(self):
        return self.net.feature_dim


class Focuser(nn.Module):
    def __init__(self, size=96, random=False, policy_params=None, focuser_base_model_params=None):
        super().__init__()
        self.net = TSN(**focuser_base_model_params)
        self.patch_size = size
        self.random = random
        self.patch_sampler = PatchSampler(self.patch_size, self.random)
        self.policy = None
        self.memory = Memory()
        if not self.random:
            assert policy_params is not None
            self._tables = {s * s: torch.from_numpy(grid_table(s)) for s in range(4, 11)}   # 16 ... 100 cells
            self.policy_action_dim = policy_params["action_dim"]
            self.ppo_continuous = policy_params["ppo_continuous"]
            pp = policy_params
            if self.ppo_continuous:
                self.policy = PPO_Continuous(pp["feature_dim"], pp["state_dim"], pp["hidden_state_dim"], pp["policy_conv"],
                                             pp["gpu"], gamma=pp["gamma"], lr=pp["policy_lr"],
                                             action_std=pp["action_std"], with_bn=pp["with_bn"])
                for pol in (self.policy.policy, self.policy.policy_old):
                    pol.frame_channels = pp.get("frame_channels")
            else:
                self.policy = PPO(pp["feature_dim"], pp["state_dim"], pp["action_dim"], pp["hidden_state_dim"],
                                  pp["policy_conv"], pp["gpu"], gamma=pp["gamma"], lr=pp["policy_lr"])

    @property
    def standard_actions_set(self):
        return self._tables

    def _get_standard_action(self, action):
        table = self._tables[self.policy_action_dim]
        if table.device != action.device:
            table = self._tables[self.policy_action_dim] = table.to(action.device)
        return table[action], None

    @torch.no_grad()
    def act(self, state, restart_batch=True):
        action = self.policy.select_action(state, self.memory, restart_batch, False)
        return action if self.ppo_continuous else self._get_standard_action(action)[0]

    def forward(self, *argv, **kwargs):
        """Returns the PATCH (STH/models/gfv_net.py:402-422), reference layout (B, 3T, P, P)."""
        if self.random:
            return self.random_patching(kwargs["input"])
        if kwargs.get("training"):
            raise NotImplementedError("training branch is out of scope")
        action = self.act(kwargs["state"], kwargs.get("restart_batch", True))
        return get_patch(kwargs["input"], action, self.patch_size)

    def random_patching(self, imgs):
        return get_patch(imgs, torch.rand(imgs.size(0), 2).to(imgs.device), self.patch_size)

    def predict(self, input):
        return self.net(input)

    @property
    def feature_dim(self):
        return self.net.feature_dim


class PatchSampler(nn.Module):
    def __init__(self, size=96, random=True):
        super().__init__()
        self.random, self.size = random, size

    def sample(self, imgs, action=None):
        if self.random:
            return self.random_sample(imgs)
        assert action is not None
        return get_patch(imgs, action, self.size)

    def random_sample(self, imgs):
        """STH/models/gfv_net.py:455-474: one crop per image at an origin drawn like STH/models/utils.py's random_crop (np.random.randint for y,
        then x; the draw and the (origin + 0.5) / (H - P) gather actions are the ActivityNet sampler's, adafocus_amd/gfv_net.py)."""
        from .gfv_net import PatchSampler as _ActSampler
        return get_patch(imgs, _ActSampler.random_actions(self, imgs), self.size)

    def forward(self, *argv, **kwargs):
        raise NotImplementedError("Policy driven patch sampler not implemented.")
