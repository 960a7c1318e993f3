"""DiffusionConditionedPolicy p_phi(pi|z) = N(mu_phi(z), Sigma_phi(z))
(active_inference_diffusion/models/policy_networks.py:12-154); forward on HIP kernels.
HierarchicalDiffusionPolicy (LSTM, never instantiated by the reference) is out of scope."""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn as nn

from .. import _native as nat
from ._holders import TableCache, lin_ln_relu_stack, params_dict


class NormalDist:
    """Minimal stand-in for torch.distributions.Normal as returned by the reference's forward: what callers read is
    entropy() (active_inference.py:359,527).  Entropies come from the HIP kernel."""

    def __init__(self, mean, std, entropy_sum):
        self.mean, self.stddev, self._entropy_sum = mean, std, entropy_sum

    def entropy(self):
        # per-dimension entropy; the kernel returns the row sum, callers only ever use .sum(-1)
        A = self.mean.shape[-1]
        return (self._entropy_sum / A).unsqueeze(-1).expand(-1, A)


class DiffusionConditionedPolicy(nn.Module):
    def __init__(self, latent_dim: int, action_dim: int, hidden_dim: int = 256, num_layers: int = 3, log_std_min: float = -20,
                 log_std_max: float = 2, use_state_dependent_std: bool = True, squash_output: bool = False):
        super().__init__()
        if not use_state_dependent_std or squash_output or num_layers != 3 or (log_std_min, log_std_max) != (-20, 2):
            raise NotImplementedError("only the configuration the reference instantiates is built (active_inference.py:84-89)")
        self.latent_dim, self.action_dim, self.hidden_dim = latent_dim, action_dim, hidden_dim
        self.log_std_min, self.log_std_max = log_std_min, log_std_max
        D = hidden_dim
        self.latent_encoder = nn.Sequential(nn.Linear(latent_dim, D), nn.LayerNorm(D), nn.ReLU(), nn.Linear(D, D))
        self.trunk = lin_ln_relu_stack([D, D, D, D])
        self.mean_head = nn.Sequential(nn.Linear(D, D // 2), nn.ReLU(), nn.Linear(D // 2, action_dim))
        self.log_std_head = nn.Sequential(nn.Linear(D, D // 2), nn.ReLU(), nn.Linear(D // 2, action_dim))
        self._initialize_weights()
        self._tab = TableCache(lambda: nat.policy_table(params_dict(self), ""))
        self._gtab = TableCache(lambda: nat.policy_table(params_dict(self, True), ""))
        self._ws = None
        self.rng_seed, self.rng_offset = 0x9011C7, 0

    def _initialize_weights(self):  # policy_networks.py:77-92
        for head in (self.mean_head, self.log_std_head):
            nn.init.orthogonal_(head[-1].weight, gain=1.0)
            nn.init.zeros_(head[-1].bias)
        for m in (self.latent_encoder, self.trunk, self.mean_head[:-1]):
            for layer in m.modules():
                if isinstance(layer, nn.Linear):
                    nn.init.xavier_uniform_(layer.weight)
                    nn.init.zeros_(layer.bias)

    def table(self):
        return self._tab.get(self)

    def grad_table(self):
        return self._gtab.get(self)

    def forward(self, z: torch.Tensor, deterministic: bool = False, eps: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor, NormalDist]:
        z = z.contiguous().float()
        if self._ws is None:
            self._ws = nat.Workspace(z.device)
        if not deterministic and eps is None:
            eps = torch.empty(z.shape[0], self.action_dim, device=z.device)
            nat.fill_normal(eps, self.rng_seed, self.rng_offset)
            self.rng_offset += (eps.numel() + 3) // 4
        action, logp, ent = nat.policy_forward(self.table(), z, None if deterministic else eps, self._ws)
        return action, logp, NormalDist(action, None, ent)

    def get_policy_entropy(self, z: torch.Tensor) -> torch.Tensor:
        _, _, dist = self.forward(z, deterministic=True)
        return dist.entropy().sum(dim=-1)
