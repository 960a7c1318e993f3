"""BeliefDynamics — Fokker-Planck Gaussian belief update (active_inference_diffusion/core/belief_dynamics.py:34-409),
diagonal-covariance path on a batched fp64 HIP kernel.

The reference's `update()` raises before returning (it differentiates a detached tensor and calls a missing
method — SURVEY Appendix A, D4), so whole-step parity is UNPINNED; the closed form implemented here is the one
its sub-functions pin: F-"gradient" = grad of the log-density `-(mu-o)/ns^2 - mu + s` (:174-210, golden-checked),
Hessian of the same density `-(1/ns^2+1) I`, mean drift `-lr*g*dt/(1+0.1|g|)` + noise, variance
`clamp(var*exp((-2h+2D)dt))` (:110-167), entropy (:368-389).  Full covariance (matrix_exp/eigh) is "next" (§8f-4).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import _native as nat


class BeliefDynamics(nn.Module):
    def __init__(self, latent_dim: int, config, n_beliefs: int = 1):
        super().__init__()
        if config.use_full_covariance:
            raise NotImplementedError("full-covariance Fokker-Planck (matrix_exp / eigh path) is not built yet")
        self.latent_dim, self.config, self.n_beliefs = latent_dim, config, n_beliefs
        self.register_buffer("mean", torch.zeros(n_beliefs, latent_dim, dtype=torch.float64))
        self.register_buffer("variance", torch.ones(n_beliefs, latent_dim, dtype=torch.float64))
        self.min_eigenvalue = max(config.min_variance, 1e-8)
        self.history = {"entropies": []}
        self.seed, self._offset = 0xF0CC, 0

    @property
    def precision(self):
        return 1.0 / self.variance

    def reset(self, initial_mean: Optional[torch.Tensor] = None, initial_cov: Optional[torch.Tensor] = None):
        self.mean.zero_(); self.variance.fill_(1.0)
        if initial_mean is not None:
            self.mean.copy_(initial_mean.to(self.mean).reshape(self.mean.shape))
        if initial_cov is not None:
            self.variance.copy_(torch.clamp(torch.diagonal(initial_cov.to(self.variance), dim1=-2, dim2=-1), min=self.min_eigenvalue).reshape(self.variance.shape))
        self.history = {"entropies": []}

    def update(self, observation: torch.Tensor, score_function: torch.Tensor, action: torch.Tensor = None,
               xi: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        c = self.config
        obs = observation.to(self.mean).reshape(self.mean.shape).contiguous()
        sc = score_function.to(self.mean).reshape(self.mean.shape).contiguous()
        if xi is None:
            x32 = torch.empty(self.mean.shape, device=self.mean.device, dtype=torch.float32)
            nat.fill_normal(x32, self.seed, self._offset); self._offset += (x32.numel() + 3) // 4
            xi = x32.double()
        ent = nat.fokker_planck_step(self.mean, self.variance, obs, sc, xi.to(self.mean).contiguous(), c.dt, c.diffusion_coefficient,
                                     c.learning_rate, c.noise_scale, c.min_variance, c.max_variance)
        self._last_entropy = ent
        return self.get_parameters()

    def get_parameters(self) -> Tuple[torch.Tensor, torch.Tensor]:
        m = self.mean.float(); v = self.variance.float()
        if self.n_beliefs == 1:
            return m[0], torch.diag(v[0])
        return m, torch.diag_embed(v)

    def entropy(self) -> torch.Tensor:
        e = 0.5 * torch.sum(torch.log(self.variance) + 2.8378770664093453, dim=-1)   # log(2 pi e)
        return e[0] if self.n_beliefs == 1 else e
