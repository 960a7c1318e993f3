"""FreeEnergyComputation (active_inference_diffusion/core/free_energy.py:12-101):
F = 1/2 mean|z - mu_p|^2/sigma_p^2 + 1/2 exp(log pi) mean|o - z|^2 + 0.01 mean|s_theta(z,t,o)|^2.
The score-network evaluation (the only non-trivial cost) is the HIP forward; the three row reductions are
`aid_free_energy_terms`-free: they reuse the generic reduction kernels through `_native`."""
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import _native as nat


class FreeEnergyComputation(nn.Module):
    def __init__(self, precision_init: float = 1.0):
        super().__init__()
        self.log_precision = nn.Parameter(torch.log(torch.tensor(float(precision_init))))

    @property
    def precision(self) -> torch.Tensor:
        return torch.exp(self.log_precision)

    def compute_loss(self, states: torch.Tensor, observations: torch.Tensor, actions: torch.Tensor, score_network: nn.Module,
                     current_time: float = 0.0, prior_mean: Optional[torch.Tensor] = None, prior_std: float = 1.0, backward: bool = False
                     ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        states, observations = states.contiguous().float(), observations.contiguous().float()
        B = states.shape[0]
        if prior_mean is None:
            prior_mean = torch.zeros_like(states)
        t = torch.full((B,), float(current_time), device=states.device)
        mask = None
        if score_network.training:      # one dropout mask for the forward and (backward=True) the reverse pass
            mask = torch.empty(B, score_network.hidden_dim, device=states.device)
            nat.fill_dropout_mask(mask, 0.1, score_network.rng_seed, score_network.rng_offset)
            score_network.rng_offset += (mask.numel() + 3) // 4
        score = score_network(states, t, observations, continuous=(0.0 <= current_time <= 1.0), drop_mask=mask)
        sq = nat.rowwise_sqdist_means(states, prior_mean.contiguous().float(), observations, score)   # [3]: mean_b sum_l (.)^2
        complexity = 0.5 * sq[0] / (prior_std ** 2)
        accuracy = -0.5 * self.precision.to(sq.device) * sq[1]
        score_reg = 0.01 * sq[2]
        fe = complexity - accuracy + score_reg
        info = {"complexity": complexity, "accuracy": -accuracy, "observation_error": sq[1], "score_regularization": score_reg,
                "precision": self.precision}
        if backward:
            # The reference returns an autograd scalar (free_energy.py:30-91); there is no graph here, so `backward=True` produces what F.backward() would:
            # dF/d(log_precision) = 1/2 exp(log pi) mean|o - z|^2 accumulated into log_precision.grad; dF/d(score-network parameters) accumulated into the
            # score network's .grad by its reverse pass (aid_score_backward, cotangent dF/dscore = 0.02 score / B); and dF/d(states), dF/d(observations)
            # returned in `info` (direct terms + the paths through the score network's inputs).
            gp = (0.5 * self.precision.to(sq.device) * sq[1]).detach().to(self.log_precision.device).reshape(self.log_precision.shape)
            self.log_precision.grad = gp if self.log_precision.grad is None else self.log_precision.grad + gp
            pi = float(self.precision)
            d_score = 0.02 * score / B
            d_obs = (-pi * (states - observations) / B).contiguous()
            _, dz = score_network.backward(states, t, observations, d_score, continuous=(0.0 <= current_time <= 1.0), drop_mask=mask, d_observation=d_obs)
            info["d_states_direct"] = ((states - prior_mean) / (prior_std ** 2) + pi * (states - observations)) / B
            info["d_states"] = info["d_states_direct"] + dz
            info["d_observations"] = d_obs
            info["d_score"] = d_score
        return fe, info

    def update_precision(self, complexity: torch.Tensor, accuracy: torch.Tensor):
        with torch.no_grad():
            err = (complexity - accuracy).to(self.log_precision.device)
            self.log_precision.data += 0.01 * err.clamp(-1, 1)
            self.log_precision.data = self.log_precision.data.clamp(-3, 3)
