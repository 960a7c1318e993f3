"""BeliefDynamics — Fokker-Planck Gaussian belief update (active_inference_diffusion/core/belief_dynamics.py:34-409) on batched fp64
HIP kernels, diagonal and full covariance.

The reference's `update()` raises before returning (it differentiates a detached tensor and calls a missing method — SURVEY Appendix A,
D4), so whole-step parity is UNPINNED; the closed form implemented here is the one its sub-functions pin: F-"gradient" = grad of the
log-density `-(mu-o)/ns^2 - mu + s` (:174-210, golden-checked), Hessian of the same density `-(1/ns^2+1) I`, mean drift
`-lr*g*dt/(1+0.1|g|)` + noise, then
  * diagonal: variance `clamp(var*exp((-2h+2D)dt))` (:154-167), entropy (:384-387);
  * full: `cov <- E cov E^T`, `E = expm((-H-H^T+2D I) dt)` (:268-294), eigen-clamp + condition-number regularisation (:296-332),
    `precision = (cov + eps I)^-1` (:334-342), entropy from logdet (:368-383).  Because that Hessian is a multiple of I every one of these
    is a spectral function of the covariance, so the belief is carried as eigenvalues + eigenvectors: one on-device symmetric
    eigen-decomposition at `reset(initial_cov=...)` (`aid_sym_eig_f64`), a diagonal update + two rebuilds per step
    (`aid_fokker_planck_full_step`).  tests/golden/fp_full.npz holds the reference's own sub-functions chained the same way.
A custom `observation_model` (a non-Gaussian likelihood) is not supported.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import _native as nat


class BeliefDynamics(nn.Module):
    def __init__(self, latent_dim: int, config, n_beliefs: int = 1):
        super().__init__()
        self.latent_dim, self.config, self.n_beliefs = latent_dim, config, n_beliefs
        self.full = bool(config.use_full_covariance)
        if self.full and latent_dim > 128:
            raise ValueError("full-covariance BeliefDynamics supports latent_dim <= 128")
        self.register_buffer("mean", torch.zeros(n_beliefs, latent_dim, dtype=torch.float64))
        if self.full:
            eye = torch.eye(latent_dim, dtype=torch.float64).expand(n_beliefs, latent_dim, latent_dim).contiguous()
            self.register_buffer("_cov", eye.clone()); self.register_buffer("_prec", eye.clone())
            self.register_buffer("_lam", torch.ones(n_beliefs, latent_dim, dtype=torch.float64)); self.register_buffer("_V", eye.clone())
        else:
            self.register_buffer("variance", torch.ones(n_beliefs, latent_dim, dtype=torch.float64))
        self.min_eigenvalue = max(config.min_variance, 1e-8)
        self.max_condition_number = 1e6
        self.history = {"entropies": [], "condition_numbers": []}
        self.seed, self._offset = 0xF0CC, 0

    # the reference exposes [L,L] / [L] buffers for its single belief; batched beliefs keep the leading dim
    @property
    def covariance(self):
        c = self._cov if self.full else torch.diag_embed(self.variance)
        return c[0] if self.n_beliefs == 1 else c

    @property
    def precision(self):
        if self.full:
            return self._prec[0] if self.n_beliefs == 1 else self._prec
        return 1.0 / self.variance

    def _kw(self):
        c = self.config
        return dict(dt=c.dt, D=c.diffusion_coefficient, lr=c.learning_rate, noise_scale=c.noise_scale, min_var=c.min_variance)

    def reset(self, initial_mean: Optional[torch.Tensor] = None, initial_cov: Optional[torch.Tensor] = None):
        self.mean.zero_()
        if initial_mean is not None:
            self.mean.copy_(initial_mean.to(self.mean).reshape(self.mean.shape))
        n, L = self.n_beliefs, self.latent_dim
        if self.full:
            eye = torch.eye(L, dtype=torch.float64, device=self.mean.device)
            self._lam.fill_(1.0); self._V.copy_(eye.expand(n, L, L)); self._cov.copy_(eye.expand(n, L, L)); self._prec.copy_(eye.expand(n, L, L))
            if initial_cov is not None:    # :80-83: stabilise the given matrix, then invert
                c0 = initial_cov.to(self.mean).reshape(n, L, L)
                for b in range(n):
                    w, V = nat.sym_eig(0.5 * (c0[b] + c0[b].T))
                    self._lam[b].copy_(w); self._V[b].copy_(V)
                cov, prec, _, cond = nat.fokker_planck_full_step(self.mean, self._lam, self._V, advance=False, **self._kw())
                self._cov.copy_(cov); self._prec.copy_(prec)
        else:
            self.variance.fill_(1.0)
            if initial_cov is not None:
                self.variance.copy_(torch.clamp(torch.diagonal(initial_cov.to(self.variance), dim1=-2, dim2=-1), min=self.min_eigenvalue).reshape(self.variance.shape))
        self.history = {"entropies": [], "condition_numbers": []}

    def update(self, observation: torch.Tensor, score_function: torch.Tensor, action: torch.Tensor = None,
               xi: Optional[torch.Tensor] = None, observation_model=None) -> Tuple[torch.Tensor, torch.Tensor]:
        if observation_model is not None:
            raise NotImplementedError("BeliefDynamics.update: only the default Gaussian observation model is built")
        obs = observation.to(self.mean).reshape(self.mean.shape).contiguous()
        sc = score_function.to(self.mean).reshape(self.mean.shape).contiguous()
        if xi is None:
            x32 = torch.empty(self.mean.shape, device=self.mean.device, dtype=torch.float32)
            nat.fill_normal(x32, self.seed, self._offset); self._offset += (x32.numel() + 3) // 4
            xi = x32.double()
        xi = xi.to(self.mean).reshape(self.mean.shape).contiguous()
        if self.full:
            cov, prec, ent, cond = nat.fokker_planck_full_step(self.mean, self._lam, self._V, obs, sc, xi, advance=True, **self._kw())
            self._cov, self._prec, self._last_cond = cov, prec, cond
        else:
            ent = nat.fokker_planck_step(self.mean, self.variance, obs, sc, xi, max_var=self.config.max_variance, **self._kw())
        self._last_entropy = ent
        return self.get_parameters()

    def get_parameters(self) -> Tuple[torch.Tensor, torch.Tensor]:
        m = self.mean.float()
        c = self._cov.float() if self.full else torch.diag_embed(self.variance.float())
        if self.n_beliefs == 1:
            return m[0], c[0]
        return m, c

    def entropy(self) -> torch.Tensor:
        lam = self._lam if self.full else self.variance
        e = 0.5 * torch.sum(torch.log(lam) + 2.8378770664093453, dim=-1)   # log(2 pi e)
        return e[0] if self.n_beliefs == 1 else e

    def get_diagnostics(self):
        """belief_dynamics.py:391-409.  Full covariance: min / max eigenvalue, condition number, determinant - read off the eigenvalues the belief is
        carried as (the reference calls eigvals / det on the matrix); diagonal: min / max / mean variance; both: |mean| and the entropy.  One belief:
        floats, as the reference; batched beliefs: one list entry per belief."""
        one = (lambda x: float(x[0])) if self.n_beliefs == 1 else (lambda x: [float(v) for v in x])
        d = {}
        if self.full:
            lam = self._lam
            d["min_eigenvalue"], d["max_eigenvalue"] = one(lam.min(dim=-1).values), one(lam.max(dim=-1).values)
            d["condition_number"] = one(lam.max(dim=-1).values / lam.min(dim=-1).values)
            d["determinant"] = one(torch.exp(torch.log(lam).sum(dim=-1)))
        else:
            v = self.variance
            d["min_variance"], d["max_variance"], d["mean_variance"] = one(v.min(dim=-1).values), one(v.max(dim=-1).values), one(v.mean(dim=-1))
        d["mean_norm"] = one(self.mean.norm(dim=-1))
        e = self.entropy()
        d["entropy"] = float(e) if self.n_beliefs == 1 else [float(x) for x in e]
        return d
