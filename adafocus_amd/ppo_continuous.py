"""Continuous patch-location policy, inference branch (STH/models/ppo_continuous.py:27-109,142-163):
encoder -> GRU step -> Linear(2)+Sigmoid; eval returns the action mean.  The producer of the (y,x)
fractions the HIP gather consumes, on the conv engine + the GRU kernel; the hidden
state is carried across ``video_div`` steps in ``memory.hidden`` like the reference does.  PPO update /
sampling are training code and absent (the reference's eval branch still draws ``dist.sample()`` and
discards it, ppo_continuous.py:98,107 -- it only advances the global RNG)."""
import torch
from torch import nn

from . import hip_ops
from .ppo import Memory  # noqa: F401  (same class in both reference files)

__all__ = ["ActorCritic", "PPO_Continuous", "Memory"]


class ActorCritic(nn.Module):
    def __init__(self, feature_dim, state_dim, hidden_state_dim=1024, policy_conv=True, action_std=0.1, with_bn=False):
        super().__init__()
        if policy_conv:
            flat = int(state_dim * 64 / feature_dim)
            if with_bn:
                self.state_encoder = nn.Sequential(nn.Conv2d(feature_dim, 64, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(),
                                                   nn.Flatten(), nn.Linear(flat, hidden_state_dim),
                                                   nn.BatchNorm1d(hidden_state_dim), nn.ReLU())
            else:
                self.state_encoder = nn.Sequential(nn.Conv2d(feature_dim, 64, 1, bias=False), nn.ReLU(), nn.Flatten(),
                                                   nn.Linear(flat, hidden_state_dim), nn.ReLU())
        else:
            self.state_encoder = nn.Sequential(nn.Linear(state_dim, 2048), nn.ReLU(), nn.Linear(2048, hidden_state_dim),
                                               nn.ReLU())
        self.gru = nn.GRU(hidden_state_dim, hidden_state_dim, batch_first=False)
        self.actor = nn.Sequential(nn.Linear(hidden_state_dim, 2), nn.Sigmoid())
        self.critic = nn.Sequential(nn.Linear(hidden_state_dim, 1))
        self.hidden_state_dim, self.policy_conv, self.feature_dim = hidden_state_dim, policy_conv, feature_dim
        self.frame_channels = None       # channels of ONE glancer frame (set by the Focuser); feature_dim = Tg * that

    @torch.no_grad()
    def act_nhwc(self, featmap_nhwc, b, tg, memory=None, restart_batch=True):
        """Clip-level action from the HIP glancer's map (B*Tg, h, w, C): the 1x1 conv over the
        channel-concatenated state (B, Tg*C, h, w) is a (Tg x 1) convolution over the (Tg, h*w) grid of
        the pixel-major map -- no concatenated tensor is built.  `memory.hidden` carries the GRU state
        across the video_div steps exactly as act() does (reset when restart_batch)."""
        if not self.policy_conv:
            raise NotImplementedError("adafocus_amd policy: policy_conv=True (the shipped configs) only")
        n, hh, ww, ch = featmap_nhwc.shape
        hw = hh * ww
        enc = self.state_encoder
        with_bn = isinstance(enc[1], nn.BatchNorm2d)
        conv, lin = enc[0], enc[4 if with_bn else 3]
        cmid = conv.weight.shape[0]
        w_enc = conv.weight.detach().view(cmid, tg, 1, ch).contiguous()
        w_lin = lin.weight.detach().view(-1, cmid, hw).permute(0, 2, 1).reshape(lin.weight.shape[0], hw * cmid).contiguous()
        sc1 = bi1 = sc2 = None
        bi2 = lin.bias.detach()
        if with_bn:
            sc1, bi1 = hip_ops.fold_bn(enc[1].weight.detach(), enc[1].bias.detach(), enc[1].running_mean, enc[1].running_var)
            sc2, t2 = hip_ops.fold_bn(enc[5].weight.detach(), enc[5].bias.detach(), enc[5].running_mean, enc[5].running_var)
            bi2 = lin.bias.detach() * sc2 + t2        # BN1d(Wx + b) = (Wx) * s + (b * s + t)
        x = featmap_nhwc.view(b, tg, hw, ch)                                   # "image" of Tg rows x hw columns
        e = hip_ops.conv2d_bn_act(x, w_enc, sc1, bi1, act=hip_ops.ACT_RELU)    # (b, 1, hw, cmid)
        e = hip_ops.conv2d_bn_act(e.view(b, 1, 1, hw * cmid), w_lin.view(-1, 1, 1, hw * cmid), sc2, bi2, act=hip_ops.ACT_RELU)
        g = self.gru
        h0 = None
        if memory is not None:
            if restart_batch:      # ppo_continuous.py:79-81: the list restarts with the zero state (k + 1 entries after k steps)
                del memory.hidden[:]
                memory.hidden.append(torch.zeros(1, b, self.hidden_state_dim, device=featmap_nhwc.device))
            if memory.hidden:
                h0 = memory.hidden[-1].view(b, -1)
        hs = hip_ops.gru_seq_forward(e.view(b, 1, -1), g.weight_ih_l0.detach(), g.weight_hh_l0.detach(),
                                     g.bias_ih_l0.detach(), g.bias_hh_l0.detach(), h0=h0)
        if memory is not None:
            memory.hidden.append(hs.view(1, b, -1))
        a = self.actor[0]
        return hip_ops.linear(hs.view(b, -1), a.weight.detach(), a.bias.detach(), act=hip_ops.ACT_SIGMOID)

    @torch.no_grad()
    def act(self, state_ini, memory, restart_batch=False, training=False):
        """Reference signature (ppo_continuous.py:78-109): state_ini (B, Tg*C, h, w), the glancer maps of one video_div
        segment concatenated on the channel axis.  Re-laid out pixel-major (glue) and run on the engine."""
        if training:
            raise NotImplementedError("adafocus_amd implements the inference branch of the policy only")
        b, tc, hh, ww = state_ini.shape
        ch = self.frame_channels or (1280 if tc % 1280 == 0 else tc)    # channels per glancer frame (feature_map_channels)
        tg = tc // ch
        nhwc = state_ini.view(b, tg, ch, hh, ww).permute(0, 1, 3, 4, 2).contiguous().view(b * tg, hh, ww, ch)
        return self.act_nhwc(nhwc, b, tg, memory, restart_batch)


class PPO_Continuous:
    """Plain holder (not an nn.Module, like the reference: its weights live under the checkpoint's
    separate 'policy' key, STH/evaluate.py:142-146)."""

    def __init__(self, feature_dim, state_dim, hidden_state_dim, policy_conv, gpu=0, lr=0.0003, betas=(0.9, 0.999),
                 gamma=0.7, K_epochs=1, eps_clip=0.2, action_std=0.1, with_bn=False):
        self.policy = ActorCritic(feature_dim, state_dim, hidden_state_dim, policy_conv, action_std, with_bn)
        self.policy_old = ActorCritic(feature_dim, state_dim, hidden_state_dim, policy_conv, action_std, with_bn)
        self.policy_old.load_state_dict(self.policy.state_dict())

    def to(self, device):
        self.policy.to(device)
        self.policy_old.to(device)
        return self

    def select_action(self, state, memory, restart_batch=False, training=True):
        return self.policy_old.act(state, memory, restart_batch, training)

    def update(self, memory):
        raise NotImplementedError("PPO update is training code")
