"""Patch-location policy, inference branch only (ACT/models/ppo.py:27-96,125-145).

The policy is the PRODUCER of the crop coordinates; its output tensor is handed
to the HIP gather without a host round trip.  The PPO update / Memory replay logic is training code
and out of scope.

``ActorCritic.act_sequence_nhwc`` is the offline-inference form on the HIP engine: in eval mode the
policy input is only the glancer feature map and its own hidden state (ppo.py:67-96), so all T
actions are computed before any patch is cropped -- 1x1 conv + Linear over all B*T frames at once,
one GRU scan, one actor GEMM, arg-max + table lookup in one small kernel.  ``act`` (one step,
reference signature, hidden state carried in ``memory.hidden`` element for element like the reference: the zero
state after `restart_batch`, then one entry per step) runs the same kernels with T = 1 and ``h0`` = the previous
step's state; both encoders (`policy_conv` True / False) run on the engine; nothing on this surface is an ATen / MIOpen op.
"""
import torch
from torch import nn

from . import hip_ops

__all__ = ["Memory", "ActorCritic", "PPO"]


class Memory:
    def __init__(self):
        self.actions, self.states, self.logprobs, self.rewards, self.is_terminals, self.hidden = [], [], [], [], [], []

    def clear_memory(self):
        for lst in (self.actions, self.states, self.logprobs, self.rewards, self.is_terminals, self.hidden):
            del lst[:]


class ActorCritic(nn.Module):
    def __init__(self, feature_dim, state_dim, action_dim, hidden_state_dim=1024, policy_conv=True):
        super().__init__()
        if policy_conv:
            self.state_encoder = nn.Sequential(
                nn.Conv2d(feature_dim, 32, 1, bias=False), nn.ReLU(), nn.Flatten(),
                nn.Linear(int(state_dim * 32 / feature_dim), hidden_state_dim), nn.ReLU())
        else:
            self.state_encoder = nn.Sequential(nn.Linear(state_dim, 2048), nn.ReLU(),
                                               nn.Linear(2048, hidden_state_dim), nn.ReLU())
        self.gru = nn.GRU(hidden_state_dim, hidden_state_dim, batch_first=False)
        self.actor = nn.Sequential(nn.Linear(hidden_state_dim, action_dim), nn.Softmax(dim=-1))
        self.critic = nn.Sequential(nn.Linear(hidden_state_dim, 1))
        self.hidden_state_dim, self.action_dim, self.policy_conv, self.feature_dim = \
            hidden_state_dim, action_dim, policy_conv, feature_dim

    def act(self, state_ini, memory, restart_batch=False, training=True):
        """One step, eval branch of ppo.py:67-96 (argmax of the actor's softmax)."""
        if training:
            raise NotImplementedError("adafocus_amd implements the inference branch of the policy only")
        b = state_ini.size(0)
        if restart_batch:
            # ppo.py:68-70: the list restarts with the zero state, so memory.hidden holds k + 1 entries after k steps
            del memory.hidden[:]
            memory.hidden.append(torch.zeros(1, b, self.hidden_state_dim, device=state_ini.device))
        e = self._encode(state_ini)
        g, actor = self.gru, self.actor[0]
        hs = hip_ops.gru_seq_forward(e.view(b, 1, -1), g.weight_ih_l0.detach(), g.weight_hh_l0.detach(),
                                     g.bias_ih_l0.detach(), g.bias_hh_l0.detach(), h0=memory.hidden[-1].view(b, -1))
        memory.hidden.append(hs.view(1, b, -1))
        logits = hip_ops.linear(hs.view(b, -1), actor.weight.detach(), actor.bias.detach())
        return hip_ops.argmax_rows(logits)

    def _encode(self, state):
        """state_encoder on the engine.  policy_conv=True (ppo.py:31-39: MobileNet / EfficientNet / RegNet feature maps):
        1x1 conv + ReLU + flatten + Linear + ReLU over (N, C, h, w) [reference layout] or (N, h, w, C) [pixel-major];
        policy_conv=False (ppo.py:40-47: ResNet / DenseNet features): `state.flatten(1)` through two Linear + ReLU."""
        n = state.shape[0]
        if not self.policy_conv:
            l0, l1 = self.state_encoder[0], self.state_encoder[2]
            e = hip_ops.linear(state.reshape(n, -1).contiguous(), l0.weight.detach(), l0.bias.detach(), act=hip_ops.ACT_RELU)
            return hip_ops.linear(e, l1.weight.detach(), l1.bias.detach(), act=hip_ops.ACT_RELU)
        # (B, C, h, w) reference layout -> pixel-major; free when `state` is a permuted view of the HIP glancer's map
        nhwc = state.permute(0, 2, 3, 1).contiguous() if state.shape[1] == self.feature_dim and state.shape[-1] != self.feature_dim else state
        hw = nhwc.shape[1] * nhwc.shape[2]
        w_enc, w_lin = self._hip_weights(hw)
        lin = self.state_encoder[3]
        e = hip_ops.conv2d_bn_act(nhwc, w_enc, act=hip_ops.ACT_RELU)
        return hip_ops.linear(e.view(n, -1), w_lin, lin.bias.detach(), act=hip_ops.ACT_RELU)

    def _hip_weights(self, hw):
        """Engine-layout views of the parameters (cached on the parameter versions)."""
        enc, lin = self.state_encoder[0], self.state_encoder[3]
        sig = tuple((q.data_ptr(), q._version) for q in (enc.weight, lin.weight))
        if getattr(self, "_hipw_sig", None) != sig:
            cmid = enc.weight.shape[0]
            w_enc = enc.weight.detach().reshape(cmid, 1, 1, -1).contiguous()
            # reference flattens (B, cmid, h, w) channel-major; the engine's map is pixel-major
            w_lin = lin.weight.detach().view(-1, cmid, hw).permute(0, 2, 1).reshape(lin.weight.shape[0], hw * cmid).contiguous()
            self._hipw, self._hipw_sig = (w_enc, w_lin), sig
        return self._hipw

    @torch.no_grad()
    def act_sequence_nhwc(self, featmap_nhwc, b, t, table):
        """featmap (B*T, h, w, C) pixel-major (the HIP glancer's output) -> (idx (B,T) int64,
        actions (B*T, 2) fp32 = table[idx])."""
        n = featmap_nhwc.shape[0]
        g, act = self.gru, self.actor[0]
        if self.policy_conv:
            e = self._encode(featmap_nhwc)                                                       # (n, 1024)
        else:   # the Linear encoder flattens the reference's (C, h, w) order
            e = self._encode(featmap_nhwc.permute(0, 3, 1, 2))
        hs = hip_ops.gru_seq_forward(e.view(b, t, -1), g.weight_ih_l0.detach(), g.weight_hh_l0.detach(),
                                     g.bias_ih_l0.detach(), g.bias_hh_l0.detach())
        logits = hip_ops.linear(hs.view(b * t, -1), act.weight.detach(), act.bias.detach())
        idx, actions = hip_ops.grid_actions(logits, table)
        return idx.view(b, t), actions


class PPO(nn.Module):
    """Holder of policy / policy_old with the reference's attribute and state-dict names."""

    def __init__(self, feature_dim, state_dim, action_dim, hidden_state_dim, policy_conv, gpu=0, lr=0.0003,
                 betas=(0.9, 0.999), gamma=0.7, K_epochs=1, eps_clip=0.2):
        super().__init__()
        self.lr, self.betas, self.gamma, self.eps_clip, self.K_epochs = lr, betas, gamma, eps_clip, K_epochs
        self.policy = ActorCritic(feature_dim, state_dim, action_dim, hidden_state_dim, policy_conv)
        self.policy_old = ActorCritic(feature_dim, state_dim, action_dim, hidden_state_dim, policy_conv)
        self.policy_old.load_state_dict(self.policy.state_dict())

    def select_action(self, state, memory, restart_batch=False, training=True):
        return self.policy_old.act(state, memory, restart_batch, training)

    def update(self, memory):
        raise NotImplementedError("PPO.update is training code (out of scope, SURVEY.md §2 row 5)")
