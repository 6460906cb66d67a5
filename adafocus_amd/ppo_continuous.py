"""Continuous patch-location policy, inference branch (STH/models/ppo_continuous.py:27-109,142-163):
encoder -> GRU step -> Linear(2)+Sigmoid; eval returns the action mean.  A PyTorch-ROCm producer of
the (y,x) fractions the HIP gather consumes.  PPO update / sampling are training
code and absent."""
import torch
from torch import nn

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

    @torch.no_grad()
    def act(self, state_ini, memory, restart_batch=False, training=False):
        if training:
            raise NotImplementedError("adafocus_amd implements the inference branch of the policy only")
        if restart_batch:
            del memory.hidden[:]
            memory.hidden.append(torch.zeros(1, state_ini.size(0), self.hidden_state_dim, device=state_ini.device))
        state = self.state_encoder(state_ini if self.policy_conv else state_ini.flatten(1))
        state, hidden = self.gru(state.view(1, state.size(0), state.size(1)), memory.hidden[-1])
        memory.hidden.append(hidden)
        return self.actor(state[0]).detach()


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
